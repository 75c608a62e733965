// back_pass_sh.hip — the backward pass of the SHARED time-invariant shape (src/backward_pass.jl:217-252 + :28-42,:64-76 with one
// fx, fu, cxx, cxu, cuu for the whole batch, no control limits; n = 10, m = 2), computed the way the data allows:
//
//   The matrix half of the recursion — Vxx_i, K_i, Quu_i (:242-247, :41-42, :69-72) — depends on (fx, fu, cxx, cxu, cuu, λ, regType)
//   only.  Trajectories with the same λ share it bit for bit, so it is computed ONCE per distinct λ ("group"), not once per
//   trajectory.  What is left per trajectory is affine in its own gradients:
//       Qu = cu_i + fu'Vx⁺,  Qx = cx_i + fx'Vx⁺,  k_i = -QuuF⁻¹Qu,  Vx_i = Qx + K'(Quu k + Qu) + Qux'k        (:240-241, :41, :64-69)
//     = one 16 x 12 matrix M_i applied to z = [Vx⁺; cu_i]:
//       rows 0-9   [Φ Γ]        Vx_i = cx_i + Γ cu_i + Φ Vx⁺,   Γ = K' - (K'Quu + Qux')QuuF⁻¹,  Φ = fx' + Γ fu'
//       rows 10-11 [Ψ -Fi]      k_i,                             Fi = QuuF⁻¹,  Ψ = -Fi fu'
//       rows 12-13 [fu' I]      Qu                               (for dV[1] += k'Qu)
//       rows 14-15 Quu [Ψ -Fi]  Quu k                            (for dV[2] += ½ k'Quu k)
//   i.e. 12 broadcast multiply-adds of a 16-lane row per trajectory-step instead of 7 matrix instructions + a gain solve.
//
// One launch, roles by arrival ticket (the first G work-groups to arrive are resident by definition, so nobody waits for a
// work-group that cannot start):
//   producer work-group (one per group): CHAIN wave — the fp64-MFMA tile recursion of back_pass_mx2.hip without its vector column,
//     records V, K, Quu, Qux per step in the LDS; BUILDER wave — ½(V + V'), M_i; PUBLISHER wave — copies finished chunks of 8 steps
//     (aligned to the absolute step index) to the group's record stream in global memory with write-through (sc1) stores, drains
//     them, then advances the group's progress word.
//   consumer work-groups (one per tile of <= 16 trajectories of a group): DMA wave — polls the progress word (relaxed, sc1), brings
//     published chunks into a 3-deep LDS ring by direct-to-LDS loads (sc1); AFFINE waves — 4 trajectories each, one 16-lane row
//     per trajectory: [cx;cu] chunks by their own direct-to-LDS loads, the 12 multiply-adds per step, Vx / k back out in whole
//     640 / 128 byte bursts; WRITER waves — hold the chunk's Vxx | K | Quu image in registers (10 LDS reads) and store it once per
//     trajectory of the tile: the broadcast write-back the API contract asks for (every trajectory gets all outputs) costs one LDS read
//     per 16 trajectories.
// The kernel is bound by the HBM write of the results (1 184 B per trajectory-step) and, while the batch is small, by the one chain
// wave per group (its ~900-cycle step is the critical path of the whole launch: the recursion is serial in time).
//
// Grouping (sh_group_kernel, one work-group): λ bit patterns through a 64-slot LDS hash table; the up to SH_GMAX most populated
// values with >= 2 trajectories become groups (counting sort -> perm, tiles -> work items); every other active trajectory is handed
// to the per-trajectory kernels through `fb_active` (the dispatcher launches them behind this kernel; with nothing to do they exit at once).
// diverge / zero-fill (:37-38, :226-229) follow the group: QuuF not positive definite at step i stops the chain, the failing chunk is
// published with zero records below the failing step, later chunks are written as zeros by the consumers.
#include <stdlib.h>
#include "ddp_internal.h"

namespace {

#include "back_pass_mx_common.h"

constexpr int SH_GMAX = 16;                         // groups per launch
constexpr int SH_TAB = 64;                          // hash slots for distinct λ values
constexpr int CH = 8;                               // steps per chunk: chunk c = steps 8c .. 8c + 7 (absolute index)
constexpr int GREC = 348;                           // doubles per step of a group's record stream: Vxx 100 | K 20 | Quu 4 | M 16 x 14
constexpr int G_K = 100, G_QUU = 120, G_M = 124, MLD = 14, G_OUT = 124;
constexpr int GCHUNK = GREC * CH;                   // 2 784 doubles = 22 272 bytes
constexpr int GPIECES = GCHUNK / 2;                 // 16-byte pieces per chunk
constexpr int CREC = 144;                           // chain -> builder record: V 100 | K 20 | Quu 4 | Qux 20
constexpr int C_K = 100, C_QUU = 120, C_QUX = 124;
constexpr int NCB = 2, NPB = 3, NSB = 3;            // chain record buffers, publish buffers (producer); record ring (consumer)
constexpr int SH_FIN = 1 << 30, SH_ABORT = 1 << 29, SH_CNT = (1 << 24) - 1;
#ifndef SH_TMAX
#define SH_TMAX 32
#endif
#ifndef SH_NWR
#define SH_NWR 6
#endif
#ifndef SH_SYM
#define SH_SYM 8                                    // the chain forms ½(V + V') itself every SH_SYM-th step (the builder wave does it for the records of the others)
#endif
#ifndef SH_RCP1
#define SH_RCP1 0                                   // 1: one Newton step on 1 / det of the chain's 2x2 gain solve instead of two (A/B)
#endif
#ifndef SH_STRAIGHT
#define SH_STRAIGHT 1                               // 1: no branch inside a chunk of the chain (a failing step is noted, the chunk runs to its end)
#endif
constexpr int TMAX = SH_TMAX;                       // trajectories per consumer tile
constexpr int NAFF = TMAX / 4, NWR = SH_NWR;        // affine waves, writer waves
constexpr int SH_THREADS = DDP_WAVE * (1 + NAFF + NWR);

// ---- LDS maps (doubles) ------------------------------------------------------------------------------------------------------
constexpr int P_TILE = 0;                                           // transpose tile + zero cells (TLD * 16 + 16)
constexpr int P_CREC = TLD * 16 + 16;                               // NCB chunks of CH records
constexpr int P_CDUMP = P_CREC + NCB * CH * CREC;                   // cells the lanes without an output write to
constexpr int P_PUB = P_CDUMP + 64 + CREC * (CH - 1) + 16;          // NPB chunks in the layout of the record stream
constexpr int P_FLAGS = P_PUB + NPB * GCHUNK;
static_assert(P_PUB % 2 == 0 && P_FLAGS % 2 == 0, "16-byte pieces");
constexpr int C_SBUF = 0;                                           // NSB chunks of the record stream
constexpr int EIMG = 4 * 96;                                        // [cx 80 | cu 16] of a chunk for the 4 trajectories of a wave
constexpr int NEI = NSB;                                            // gradient images per affine wave: freed with the record ring's slots
constexpr int C_WAVE = NSB * GCHUNK, C_WSZ = NEI * EIMG + 8;        // per affine wave: NEI gradient images (the results overwrite the gradients in place), zero cell, dump cell
constexpr int C_FLAGS = C_WAVE + NAFF * C_WSZ;
constexpr int SH_LDS_DOUBLES = (P_FLAGS > C_FLAGS ? P_FLAGS : C_FLAGS) + 48;
constexpr size_t SH_LDS_BYTES = (size_t)SH_LDS_DOUBLES * 8;
static_assert(SH_LDS_BYTES <= 160 * 1024, "LDS budget");
// producer flags (ints behind P_FLAGS): chain -> builder -> publisher
enum { PF_CREADY = 0, PF_BDONE = 1, PF_BREADY = 2, PF_PDONE = 3, PF_CDIV = 4 };
// consumer flags (ints behind C_FLAGS)
enum { CF_SREADY = 0, CF_KIND = 1 /* NSB */, CF_UDONE = 4 /* NAFF + NWR */ };
static_assert(CF_UDONE + NAFF + NWR <= 32 && SH_THREADS <= 1024, "flag words, work-group size");

struct ShCtl {                                      // device-resident control block of a launch
    int ticket, G, W, nfb;
    int error, errors_total, pad1, pad2;              // error: timed-out tiles of this launch; errors_total: since the block was allocated
    int progress[SH_GMAX * 16];                     // chunks published by group g (| SH_FIN) at [16 g]: one 64-byte line each
    int gdiverge[SH_GMAX];
    int gcount[SH_GMAX], gstart[SH_GMAX];
    double glam[SH_GMAX];
    unsigned long long prof[64];                    // -DSH_PROF: phase sums of group 0's chain wave + wall-clock marks (ddp_sh_prof)
    // the first SH_NDIAG tiles that gave up since the block was allocated (ddp_sh_timeout_info): which tile of which group waited for
    // which chunk, the progress word it last saw, for how long, on which XCD, with which ticket, in which launch
    int ndiag, launches, pad3, pad4;
    int diag[8][8];                                 // {tile, group, chunk, progress word seen, waited ms, xcc id, G of the launch, launch number}
};
constexpr int SH_NDIAG = 8;

struct ShArgs {
    int N, B, ncu, regType, wmax, test_abort;                   // wmax: the items[] capacity = the work-groups behind the SH_GMAX producers of the grid
    const double *cx, *cu, *cxx, *cxu, *cuu, *fx, *fu, *lambda;
    const int32_t *active;
    double *K, *k, *Quu, *Vx, *Vxx, *dV;
    int32_t *diverge;
    ShCtl *ctl;
    int4 *items;                                    // (group, first position in perm, trajectories, 0)
    int *perm;
    int32_t *fb_active;
    double *sink;                                   // 4 KB that lanes without a result may write (the handle's)
    double *rec;                                    // record streams: [SH_GMAX][chunks][CH][GREC]
};

typedef __attribute__((address_space(3))) int lds_int;
__device__ __forceinline__ int lds_load_flag(const int *p) { return *(const volatile lds_int *)p; }
__device__ __forceinline__ void lds_store_flag(int *p, int v) { asm volatile("" ::: "memory"); *(volatile lds_int *)p = v; asm volatile("" ::: "memory"); }

__device__ __forceinline__ void store16_sc1(void *p, d2 v)
{
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void store4_sc1(int *p, int v)
{
    asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ int load4_sc1(const int *p)
{
    int v;
    asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
// (a store of more than 8 bytes reads its data registers AFTER it has been issued: the next instruction must not overwrite them — a
// hazard the compiler pads for its own stores and cannot see inside an asm statement.  Round 5 met it: with the address select in front
// of each store the allocator reused the data registers at once and the LOW DWORDS of some stored k / Vx were those of the next value.)
// NT: non-temporal (the lines do not linger in the L2) or plain.  Measured (profiles/r05_sh_stores.txt): the same at B = 1 024-2 048, plain
// 12 % faster at B = 32 768 (6.57 vs 7.45 ms: the write-back of full lines from the L2 is what the HBM likes); write-through (sc1) stores
// triple the time of the small launches.
template <bool NT>
__device__ __forceinline__ void store16_res(void *p, d2 v)
{
    if (NT) asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

// =============================================================== grouping ====================================================
// REGS: every thread keeps the table slots of its (at most 8) trajectories in registers between the two passes; otherwise they are parked
// in global memory (perm[], then fb_active[]).  The kernel is a chain of dependent steps on ONE work-group in front of the whole backward
// pass: every global round trip in it (~1-2 us) is time the 256 CUs wait.
template <bool REGS>
__global__ __launch_bounds__(1024) void sh_group_kernel(ShArgs a)
{
    __shared__ unsigned long long keys[SH_TAB];
    __shared__ int cnt[SH_TAB], gof[SH_TAB], cursor[SH_GMAX], gst[SH_GMAX], gcn[SH_GMAX], gtile[SH_GMAX + 1], Gs, Ts, nfb;
    const unsigned long long EMPTY = ~0ull;
    const int tid = threadIdx.x, B = a.B, lane = tid & 63;
    constexpr int NPT = 8;
    int slotr[NPT];
    if (tid < SH_TAB) { keys[tid] = EMPTY; cnt[tid] = 0; gof[tid] = -1; }
    if (tid < SH_GMAX) cursor[tid] = 0;
    if (tid == 0) nfb = 0;
    __syncthreads();
    // pass 1: distinct values and their populations.  A wave whose active lanes all carry the same λ (the usual case: one λ for the
    // batch, or a few values in runs) inserts ONCE and adds its population once — per-lane atomics on one LDS word serialise (18 us at
    // B = 1 024, 158 us at B = 32 768 before this).
    auto insert = [&](unsigned long long key, int n_) -> int {
        unsigned h = (unsigned)((key * 0x9E3779B97F4A7C15ull) >> 58);
        for (int t = 0; t < SH_TAB; ++t, h = (h + 1) % SH_TAB) {
            const unsigned long long old = atomicCAS(&keys[h], EMPTY, key);
            if (old == EMPTY || old == key) { atomicAdd(&cnt[h], n_); return (int)h; }
        }
        return -2;
    };
    auto pass1 = [&](int b) -> int {
        const bool inb = b < B;
        const bool on = inb && (!a.active || a.active[b] != 0);
        const unsigned long long key = on ? (unsigned long long)__double_as_longlong(a.lambda[b]) : EMPTY;
        const unsigned long long live = __ballot(on && key != EMPTY);
        int slot = on ? -2 : -1;
        // peel the wave value by value: the lanes that share the leader's λ insert once, with their head count
        unsigned long long rest = live;
        while (rest) {
            const int lead = __ffsll((long long)rest) - 1;
            const unsigned long long k0 = __shfl(key, lead);
            const unsigned long long same = __ballot(on && key == k0) & rest;
            int s0 = 0;
            if (lane == lead) s0 = insert(k0, __popcll(same));
            s0 = __shfl(s0, lead);
            if ((same >> lane) & 1ull) slot = s0;
            rest &= ~same;
        }
        return slot;                                 // -1 inactive, -2 no slot, else the slot
    };
    if (REGS) {
#pragma unroll
        for (int q = 0; q < NPT; ++q) slotr[q] = (q * 1024 < B) ? pass1(q * 1024 + tid) : -1;      // (block-uniform condition)
    } else {
        for (int b0 = 0; b0 < B; b0 += blockDim.x) { const int sl = pass1(b0 + tid); if (b0 + tid < B) a.perm[b0 + tid] = sl; }
    }
    __syncthreads();
    if (tid < 64) {   // the SH_GMAX most populated values with at least two trajectories become groups (wave 0: an arg-max per group)
        int mine = cnt[tid] > 1 ? (cnt[tid] << 8) | (SH_TAB - 1 - tid) : 0;      // ties: the lowest slot
        int G = 0, start = 0;
        for (; G < SH_GMAX; ++G) {
            int best = mine;
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) { const int o = __shfl_xor(best, off, 64); best = o > best ? o : best; }
            if (best == 0) break;
            const int sb = SH_TAB - 1 - (best & 0xff), bc = best >> 8;
            if (tid == sb) { mine = 0; gof[sb] = G; }
            if (tid == 0) {
                gst[G] = start; gcn[G] = bc;
                a.ctl->glam[G] = __longlong_as_double((long long)keys[sb]);
                a.ctl->gcount[G] = bc; a.ctl->gstart[G] = start;
            }
            start += bc;
        }
        // Tile size: one work-group per CU is resident (LDS), the G producers hold a CU each while the chain runs.  R rounds of
        // (ncu - G) tiles of equal size, no tile more than TMAX trajectories: every CU is busy until the end, nobody queues behind a
        // full machine for a lone last tile.  (The host sizes items[] and the grid for the most tiles ANY grouping of the batch can
        // produce — ddp_sh_max_tiles — and the count is clamped to that capacity here.)
        if (tid == 0) {
            int T = 4, W = 0;
            if (G > 0) {
                const int slots = a.ncu - G > 8 ? a.ncu - G : 8;
                const int R = (start + slots * TMAX - 1) / (slots * TMAX);
                T = (start + R * slots - 1) / (R * slots);
                T = T < 4 ? 4 : T;
                // never more tiles than the host sized items[] and the grid for (at T = TMAX the count is <= B / TMAX + G <= wmax)
                const int cap = R * slots < a.wmax ? R * slots : a.wmax;
                for (; T < TMAX; ++T) { int w = 0; for (int g = 0; g < G; ++g) w += (gcn[g] + T - 1) / T; if (w <= cap) break; }
            }
            for (int g = 0; g < G; ++g) { gtile[g] = W; W += (gcn[g] + T - 1) / T; }
            gtile[G] = W;
            Gs = G; Ts = T;
            a.ctl->G = G; a.ctl->W = W; a.ctl->ticket = 0; a.ctl->error = 0; a.ctl->launches += 1;
#ifdef SH_PROF
            for (int e = 0; e < 64; ++e) a.ctl->prof[e] = 0;
            a.ctl->prof[19] = ~0ull;
#endif
        }
    }
    if (tid < SH_GMAX) { a.ctl->progress[16 * tid] = 0; a.ctl->gdiverge[tid] = 0; }
    __syncthreads();
    {   // the tiles, one per thread
        const int G = Gs, T = Ts, W = gtile[G];
        for (int w = tid; w < W; w += blockDim.x) {
            int g = 0;
            while (g + 1 < G && gtile[g + 1] <= w) ++g;
            const int t0 = (w - gtile[g]) * T;
            a.items[w] = make_int4(g, gst[g] + t0, gcn[g] - t0 < T ? gcn[g] - t0 : T, 0);
        }
    }
    // pass 2: counting sort into perm
    auto pass2 = [&](int b, int slot) {
        const bool inb = b < B;
        const int g = slot >= 0 ? gof[slot] : -1;
        const bool fbk = inb && g < 0 && slot != -1;
        unsigned long long rest = __ballot(g >= 0);
        while (rest) {                                          // one atomic per wave and group present in it
            const int lead = __ffsll((long long)rest) - 1;
            const int g0 = __shfl(g, lead);
            const unsigned long long same = __ballot(g == g0) & rest;
            int base = 0;
            if (lane == lead) base = atomicAdd(&cursor[g0], __popcll(same));
            base = __shfl(base, lead);
            if ((same >> lane) & 1ull) a.perm[gst[g0] + base + __popcll(same & ((1ull << lane) - 1ull))] = b;
            rest &= ~same;
        }
        const unsigned long long fbm = __ballot(fbk);
        if (fbm && lane == __ffsll((long long)fbm) - 1) atomicAdd(&nfb, __popcll(fbm));
        if (inb) a.fb_active[b] = fbk ? 1 : 0;
    };
    if (REGS) {
#pragma unroll
        for (int q = 0; q < NPT; ++q) if (q * 1024 < B) pass2(q * 1024 + tid, slotr[q]);
    } else {
        // (the slot of b was parked at perm[b]; positions >= the sorted prefix are only read, never written, before their own thread has
        // picked them up — the sorted area may overlap unread slots, so park them in fb_active first)
        for (int b = tid; b < B; b += blockDim.x) a.fb_active[b] = a.perm[b];
        __syncthreads();
        for (int b0 = 0; b0 < B; b0 += blockDim.x) { const int b = b0 + tid; pass2(b, b < B ? a.fb_active[b] : -1); }
    }
    __syncthreads();
    if (tid == 0) a.ctl->nfb = nfb;
}

// the 64-bit time base (100 MHz) bounds every cross-work-group wait: a protocol error ends the launch with ctl->error set
__device__ __forceinline__ bool timed_out(unsigned long long t0) { return wall_clock64() - t0 > 400000000ull; }   // 4 s

// ============================================================= the producer ==================================================
template <bool REG2>
__device__ __forceinline__ void sh_chain(const ShArgs &a, double *sm, const int gidx, const double lam)
{
    const int lane = threadIdx.x % DDP_WAVE, l15 = lane & 15, l4 = lane >> 4;
    const int N = a.N;
    int *flags = (int *)(sm + P_FLAGS);
    double *lds = sm + P_TILE, *crec = sm + P_CREC;
    __builtin_amdgcn_s_setprio(3);
    const double *fx = a.fx, *fu = a.fu, *cxx = a.cxx, *cxu = a.cxu, *cuu = a.cuu;
    auto Hel = [&](int row, int col) -> double {             // H = [cxx cxu; cxu' cuu] (p x p), zero outside
        if (col < p && row < p) {
            if (row < n && col < n) return cxx[row + n * col];
            if (row < n) return cxu[row + n * (col - n)];
            if (col < n) return cxu[col + n * (row - n)];
            return cuu[(row - n) + m * (col - n)];
        }
        return 0.0;
    };
    auto Fel = [&](int row, int col) -> double {             // F = [fx fu] (n x p), zero outside
        if (row < n && col < n) return fx[row + n * col];
        if (row < n && col < p) return fu[row + n * (col - n)];
        return 0.0;
    };
    const int urow = n + (l4 & 1);
    const double hmask = l15 < p ? 0.5 : 0.0, fmask = l15 < p ? 1.0 : 0.0;
    double F[3], Fh[3], Ff[3], Hc[4], S[3];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        Hc[s] = Hel(s < 3 ? l4 + 4 * s : urow, l15);
        if (s < 3) { F[s] = Fel(l4 + 4 * s, l15 < p ? l15 : n + (l15 & 1)); Fh[s] = hmask * F[s]; Ff[s] = fmask * F[s]; }
    }
    const d4 Hc4 = d4{Hc[0], Hc[1], Hc[2], Hc[3]};
#pragma unroll
    for (int s = 0; s < 3; ++s) { const int row = l4 + 4 * s; S[s] = (l15 < n && row < n) ? 2.0 * cxx[row + n * l15] : 0.0; }   // 2 Vxx_N (:234)
    for (int e = lane; e < TLD * 16 + 16; e += DDP_WAVE) lds[e] = 0.0;
    const double cB = l4 >= 2 ? 1.0 : -lam;                   // regType 1: T = -λK in the 16-lane rows 0, 1
    const double chl = -0.5 * lam;
    const bool odd = (l4 & 1) != 0, hi2 = l4 >= 2;
    const int wr = l4 + TLD * l15, rdT = l15 + TLD * l4, rdS = 4 * TLD;
    // where this lane's values go in a step record (lanes without one aim behind the records)
    const int DUMP = NCB * CH * CREC + lane;
    const bool r1 = l15 < n, r2 = l15 < n, r3 = !hi2 && l15 < p;
    const int w1 = r1 ? l4 + n * l15 : DUMP;
    const int w2 = r2 ? (!hi2 ? l4 + 8 + n * l15 : C_K + (l4 - 2) + m * l15) : DUMP;
    const int w3 = r3 ? (l15 < n ? C_QUX + l4 + m * l15 : C_QUU + l4 + m * (l15 - n)) : DUMP;
    const d4 zero4 = d4{0.0, 0.0, 0.0, 0.0};
    wave_sync();
    int diverge = 0;
    int wb1 = w1, wb2 = w2, wb3 = w3;
    // One time step; SIN: S holds V + V' (else V) of the step before; SOUT: ½(V + V') on the chain (else the builder does it)
#ifdef SH_PROF
    // phase profile of the chain wave (s_memtime, shader clock): [SOUT][phase] sums over all steps; phases: 0 the six products up to the
    // first use of G, 1 the 2x2 gain solve up to the issue of the update product, 2 the update product up to V in S, 3 records + branches
    unsigned long long pacc[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}}, pcnt[2] = {0, 0}, pt[5];
    const unsigned long long wc0 = wall_clock64();
// (the stamp takes the value it follows as an operand so that the compiler cannot move it; the wait makes the SGPR pair valid)
#define SH_PT(k, val) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(pt[k]), "+v"(val))
#else
#define SH_PT(k, val)
#endif
    auto step = [&](const int i, const int roff, auto sin_c, auto sout_c) __attribute__((always_inline)) {
        constexpr bool SIN = decltype(sin_c)::value != 0, SOUT = decltype(sout_c)::value != 0;
        const double *Bg = SIN ? Fh : Ff;
#ifdef SH_PROF
        SH_PT(0, S[0]);
#endif
        // GEMM1: W = Vxx·F
        d4 w = __builtin_amdgcn_mfma_f64_16x16x4f64(S[0], Bg[0], zero4, 0, 0, 0);
        w = __builtin_amdgcn_mfma_f64_16x16x4f64(S[1], Bg[1], w, 0, 0, 0);
        w = __builtin_amdgcn_mfma_f64_16x16x4f64(S[2], Bg[2], w, 0, 0, 0);
        // GEMM2: G = F'W + H  (:242-244)
        d4 g = __builtin_amdgcn_mfma_f64_16x16x4f64(F[0], w.x, Hc4, 0, 0, 0);
        g = __builtin_amdgcn_mfma_f64_16x16x4f64(F[1], w.y, g, 0, 0, 0);
        g = __builtin_amdgcn_mfma_f64_16x16x4f64(F[2], w.z, g, 0, 0, 0);
        double Z = g.w + 0.0;                              // G row 10 | 11 (Qux | Quu) with the parity of my 16-lane row
        const double hZ = 0.5 * Z;
        SH_PT(1, Z);
        double Q0, Q1, F00, F01, F11;
        if (REG2) {                                        // u-rows of F'(W + λF) + H: Qux_reg, QuuF  (:245-247)
            const double lamB = SIN ? 2.0 * lam : lam;
            d4 gr = __builtin_amdgcn_mfma_f64_16x16x4f64(F[0], fma(lamB, Bg[0], w.x), Hc4, 0, 0, 0);
            gr = __builtin_amdgcn_mfma_f64_16x16x4f64(F[1], fma(lamB, Bg[1], w.y), gr, 0, 0, 0);
            gr = __builtin_amdgcn_mfma_f64_16x16x4f64(F[2], fma(lamB, Bg[2], w.z), gr, 0, 0, 0);
            const double Zr = gr.w + 0.0;
            spread_pair(Zr, Q0, Q1);
            F00 = row_bcast<n>(Q0); F01 = row_bcast<n + 1>(Q0); F11 = row_bcast<n + 1>(Q1);
        } else {
            spread_pair(Z, Q0, Q1);
            F00 = row_bcast<n>(Q0) + lam; F01 = row_bcast<n + 1>(Q0); F11 = row_bcast<n + 1>(Q1) + lam;
        }
        const double det = fma(F00, F11, -(F01 * F01));
        const double n0 = fma(F11, Q0, -(F01 * Q1)), n1 = fma(F00, Q1, -(F01 * Q0));
#if SH_RCP1
        // 1 / det by ONE Newton step folded into the product: K = -(n y0)(2 - det y0), relative error (2^-24.4)^2 = 2^-48.7 (profiles/
        // microbench/rcp_f64_accuracy.hip) — the dependent chain behind v_rcp_f64 is mul | fma, mul (2 levels) instead of 4 fma + mul
        const double y0 = __builtin_amdgcn_rcp(det);
        const double e2 = fma(-det, y0, 2.0);
        const double Ksel = -(((odd ? n1 : n0) * y0) * e2);
        const double y = y0 * e2;
        const double K0 = -(n0 * y), K1 = -(n1 * y);       // (regType 2 only: the compiler drops them otherwise)
#else
        const double y = rcp_nr(det);
        const double K0 = -(n0 * y), K1 = -(n1 * y);       // K = -QuuF⁻¹ Qux_reg  (:42)
        const double Ksel = odd ? K1 : K0;
#endif
        double Tsel, Bop;
        // value update (:69-72): V = G + K'T + Qux'K with T = Quu K + Qux: the rank-4 product [K' Qux'][T; K].
        double Aop;
        if (!REG2 && SH_STRAIGHT) {
            // regType 1: T = -λK and Qux'K = -Qux'ΦQux is symmetric (Φ = QuuF⁻¹ is, and Qux_reg = Qux), so Qux'K = K'Qux and
            // V = G + K'Y, Y = T + Qux: the A operand is K' in all four k slots (slot k holds row k & 1), the B operand ½Y — no per-row
            // selects on the way to the product; ½(V + V') (:71-72) removes the rounding-level difference between the two forms.
            // (Not for regType 2: K = -ΦQux_reg with Qux_reg != Qux, Qux'K is not symmetric.)
            Bop = fma(chl, Ksel, hZ);                      // ½(Qux - λK)
            Aop = Ksel;
        } else if (!REG2) {
            Bop = Ksel * cB;                               // T = -λK (rows 0, 1) | K (rows 2, 3)
            Aop = hi2 ? Z : Ksel;
        } else {
            Tsel = Z;
            fmac_bcast<n, 0xf, true>(Tsel, Z, K0);
            fmac_bcast<n + 1>(Tsel, Z, K1);
            Bop = hi2 ? Ksel : Tsel;
            Aop = hi2 ? Z : Ksel;
        }
        SH_PT(2, Bop);
        const d4 v = __builtin_amdgcn_mfma_f64_16x16x4f64(Aop, Bop, g, 0, 0, 0);
        const bool badu = (__builtin_amdgcn_ballot_w64(!(F00 > 0.0)) | __builtin_amdgcn_ballot_w64(!(det > 0.0))) != 0;
        crec[wb3 + roff] = Z;
#if SH_STRAIGHT
        // diverge = i (:37-38), noted without a branch: the steps the chain still runs in this chunk work on garbage that nobody reads
        // (the builder zero-fills the failing step and everything below it, the chain stops at the end of the chunk)
        { const int bd = badu ? i + 1 : 0; diverge = diverge ? diverge : bd; }
#else
        if (__builtin_expect(badu, 0)) { diverge = i + 1; return; }     // diverge = i (:37-38)
#endif
        if (SOUT) {
            lds[wr] = v.x; lds[wr + 4] = v.y; lds[wr + 8] = v.z;
            wave_sync();
            S[0] = v.x + lds[rdT]; S[1] = v.y + lds[rdT + rdS]; S[2] = v.z + lds[rdT + 2 * rdS];
            crec[wb1 + roff] = 0.5 * S[0];
            crec[wb1 + 4 + roff] = 0.5 * S[1];
            crec[wb2 + roff] = hi2 ? Ksel : 0.5 * S[2];
            wave_sync();
        } else {
#ifdef SH_PROF
            S[0] = v.x + 0.0; S[1] = v.y; S[2] = v.z;        // (a real VALU read of V: the stamp behind it is taken when V has arrived)
#else
            S[0] = v.x; S[1] = v.y; S[2] = v.z;
#endif
#ifdef SH_PROF
            SH_PT(3, S[0]);
#endif
            crec[wb1 + roff] = S[0];
            crec[wb1 + 4 + roff] = S[1];
            crec[wb2 + roff] = hi2 ? Ksel : S[2];
        }
#ifdef SH_PROF
        if (SOUT) pt[3] = pt[2];                          // (SOUT steps: phases 2 + 3 together in [1][3])
        SH_PT(4, S[2]);
        for (int e = 0; e < 4; ++e) pacc[SOUT][e] += pt[e + 1] - pt[e];
        ++pcnt[SOUT];
#endif
    };
    const int cTop = (N - 1) / CH, st = (N - 1) % CH, NCHK = cTop + 1;
    // top chunk: the steps below the terminal one, always symmetrised on the chain
    {
        wb1 = w1; wb2 = w2; wb3 = w3;                            // buffer 0
        for (int slot = st - 1; slot >= 0 && diverge == 0; --slot) step(CH * cTop + slot, CREC * slot, IC<1>{}, IC<1>{});
        if (diverge) { lds_store_flag(&flags[PF_CDIV], diverge); lds_store_flag(&flags[PF_CREADY], 1 | SH_FIN); return; }
        lds_store_flag(&flags[PF_CREADY], NCHK == 1 ? (1 | SH_FIN) : 1);
    }
    int seen = 0;
    for (int q = 1; q < NCHK && diverge == 0; ++q) {
        while (__builtin_expect(seen < q - 1, 0)) seen = lds_load_flag(&flags[PF_BDONE]);      // buffer q % 2 was chunk q - 2
        const int boff = (q % NCB) * CH * CREC;
        wb1 = r1 ? w1 + boff : w1; wb2 = r2 ? w2 + boff : w2; wb3 = r3 ? w3 + boff : w3;
        const int ib = CH * (cTop - q);
        static_for<0, CH>([&](auto sc) __attribute__((always_inline)) {
            constexpr int slot = CH - 1 - decltype(sc)::value;
            constexpr int SINc = (slot == CH - 1 || (slot + 1) % SH_SYM == 0) ? 1 : 0, SOUTc = slot % SH_SYM == 0 ? 1 : 0;
#if SH_STRAIGHT
            step(ib + slot, CREC * slot, IC<SINc>{}, IC<SOUTc>{});
            __builtin_amdgcn_sched_barrier(0);                // (without it the scheduler parks the records of all 8 steps in registers and spills)
#else
            if (diverge == 0) step(ib + slot, CREC * slot, IC<SINc>{}, IC<SOUTc>{});
#endif
            if constexpr (slot == CH / 2) seen = lds_load_flag(&flags[PF_BDONE]);
        });
        if (diverge) { lds_store_flag(&flags[PF_CDIV], diverge); lds_store_flag(&flags[PF_CREADY], (q + 1) | SH_FIN); return; }
        lds_store_flag(&flags[PF_CREADY], q + 1);
#ifdef SH_PROF
        if (gidx == 0 && lane == 0 && q >= NCHK - 4) a.ctl->prof[40 + q - (NCHK - 4)] = wall_clock64();
#endif
    }
    if (NCHK > 1) lds_store_flag(&flags[PF_CREADY], NCHK | SH_FIN);
#ifdef SH_PROF
    if (gidx == 0 && lane == 0) {
        for (int so = 0; so < 2; ++so) { for (int e = 0; e < 4; ++e) a.ctl->prof[4 * so + e] = pacc[so][e]; a.ctl->prof[8 + so] = pcnt[so]; }
        a.ctl->prof[16] = wc0; a.ctl->prof[17] = wall_clock64();
    }
#endif
}
#ifdef SH_PROF
#define SH_MARK_MAX(slot) do { if (threadIdx.x % DDP_WAVE == 0) atomicMax(&a.ctl->prof[slot], wall_clock64()); } while (0)
#define SH_MARK_MIN(slot) do { if (threadIdx.x % DDP_WAVE == 0) { if (a.ctl->prof[slot] == 0) a.ctl->prof[slot] = ~0ull >> 1; atomicMin(&a.ctl->prof[slot], wall_clock64()); } } while (0)
#else
#define SH_MARK_MAX(slot)
#define SH_MARK_MIN(slot)
#endif
// time line of the last four chunks (group 0 / its first tile): prof[base + q - (NCHK - 4)] = wall clock
#ifdef SH_PROF
#define SH_TL(base, q, nchk, on) do { if ((on) && threadIdx.x % DDP_WAVE == 0 && (q) >= (nchk) - 4 && (q) < (nchk)) a.ctl->prof[(base) + (q) - ((nchk) - 4)] = wall_clock64(); } while (0)
#else
#define SH_TL(base, q, nchk, on)
#endif

template <bool REG2>
__device__ __forceinline__ void sh_builder(const ShArgs &a, double *sm, const double lam)
{
    const int lane = threadIdx.x % DDP_WAVE;
    const int N = a.N;
    int *flags = (int *)(sm + P_FLAGS);
    const double *crec = sm + P_CREC;
    double *pub = sm + P_PUB;
    const double *fx = a.fx, *fu = a.fu;
    // M entries of this lane: row j, columns tq, tq + 4, tq + 8
    const int j = lane & 15, tq = lane >> 4, jj = j < n ? j : n - 1, pp = j & 1;
    const int cls = j < n ? 0 : (j < n + 2 ? 1 : (j < n + 4 ? 2 : 3));
    double c0[3], b0[3], b1[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int l = tq + 4 * t;
        b0[t] = l < n ? fu[l] : (l == n ? 1.0 : 0.0);
        b1[t] = l < n ? fu[l + n] : (l == n + 1 ? 1.0 : 0.0);
        c0[t] = cls == 0 ? (l < n ? fx[l + n * j] : 0.0) : (cls == 2 ? (l < n ? fu[l + n * pp] : (l - n == pp ? 1.0 : 0.0)) : 0.0);
    }
    double btb00 = 1.0, btb01 = 0.0, btb11 = 1.0;             // regType 2: QuuF = Quu + λ fu'fu  (:245-247)
    if (REG2) {
        btb00 = btb01 = btb11 = 0.0;
        for (int l = 0; l < n; ++l) { btb00 = fma(fu[l], fu[l], btb00); btb01 = fma(fu[l], fu[l + n], btb01); btb11 = fma(fu[l + n], fu[l + n], btb11); }
    }
    // the pair (e0, e0 + 1) of Vxx this lane symmetrises and its transposed partners
    const int e0 = 2 * (lane < 50 ? lane : 0), pi = e0 % n, pj = e0 / n, tp0 = pj + n * pi, tp1 = pj + n * (pi + 1);
    const int cTop = (N - 1) / CH, st = (N - 1) % CH, NCHK = cTop + 1;
    int pdone = 0;
    for (int q = 0; q < NCHK; ++q) {
        int r;
        for (;;) {
            r = lds_load_flag(&flags[PF_CREADY]);
            if ((r & SH_CNT) > q || (r & SH_FIN)) break;
            __builtin_amdgcn_s_sleep(2);
        }
        const int have = r & SH_CNT;
        if (have <= q) break;                                  // the chain stopped before this chunk
        while (pdone < q - (NPB - 1)) { pdone = lds_load_flag(&flags[PF_PDONE]); if (pdone < q - (NPB - 1)) __builtin_amdgcn_s_sleep(4); }
        asm volatile("" ::: "memory");
        const bool last = (r & SH_FIN) && have == q + 1;
        const int div = last ? lds_load_flag(&flags[PF_CDIV]) : 0;      // 1-based failing step (0: none)
        const int sfail = div ? (div - 1) - CH * (cTop - q) : -1;        // its slot in this chunk
        const double *rc = crec + (q % NCB) * CH * CREC;
        double *pb = pub + (q % NPB) * GCHUNK;
        const int hi = q == 0 ? st : CH - 1;
        for (int slot = hi; slot >= 0; --slot) {
            const double *rec = rc + CREC * slot;
            double *out = pb + GREC * slot;
            if (q == 0 && slot == st) {                        // terminal step: Vxx_N = cxx, K_N = 0, Quu_N = cuu, no map (:234-236)
                if (lane < 50) *(d2 *)(out + 2 * lane) = d2{a.cxx[2 * lane], a.cxx[2 * lane + 1]};
                else if (lane < 60) *(d2 *)(out + 2 * lane) = d2{0.0, 0.0};
                else if (lane < 62) *(d2 *)(out + 2 * lane) = d2{a.cuu[2 * (lane - 60)], a.cuu[2 * (lane - 60) + 1]};
#pragma unroll
                for (int t = 0; t < 3; ++t) out[G_M + MLD * j + tq + 4 * t] = 0.0;
                continue;
            }
            if (slot <= sfail) {                               // the failing step and the steps below it: zeros (Quu of the failing step stays)
                if (lane < 60) *(d2 *)(out + 2 * lane) = d2{0.0, 0.0};
                else if (lane < 62) *(d2 *)(out + 2 * lane) = slot == sfail ? *(const d2 *)(rec + 2 * lane) : d2{0.0, 0.0};
#pragma unroll
                for (int t = 0; t < 3; ++t) out[G_M + MLD * j + tq + 4 * t] = 0.0;
                continue;
            }
            const bool needsym = q > 0 && (slot % SH_SYM) != 0;
            if (lane < 62) {
                d2 v = *(const d2 *)(rec + 2 * lane);
                if (lane < 50 && needsym) { v.x = 0.5 * (v.x + rec[tp0]); v.y = 0.5 * (v.y + rec[tp1]); }        // :71-72
                *(d2 *)(out + 2 * lane) = v;
            }
            const double q00 = rec[C_QUU], q10 = rec[C_QUU + 1], q01 = rec[C_QUU + 2], q11 = rec[C_QUU + 3];
            const double F00 = fma(lam, btb00, q00), F01 = REG2 ? fma(lam, btb01, q01) : q01, F11 = fma(lam, btb11, q11);
            const double det = fma(F00, F11, -(F01 * F01));
            const double y = rcp_nr(det);
            const double Fi00 = F11 * y, Fi01 = -(F01 * y), Fi11 = F00 * y;
            const double K0j = rec[C_K + m * jj], K1j = rec[C_K + 1 + m * jj], X0q = rec[C_QUX + m * jj], X1q = rec[C_QUX + 1 + m * jj];
            const double X0 = fma(K0j, q00, fma(K1j, q10, X0q)), X1 = fma(K0j, q01, fma(K1j, q11, X1q));
            const double G0 = K0j - fma(X0, Fi00, X1 * Fi01), G1 = K1j - fma(X0, Fi01, X1 * Fi11);
            const double fa = pp ? Fi01 : Fi00, fb = pp ? Fi11 : Fi01;
            const double qa = pp ? q10 : q00, qb = pp ? q11 : q01;
            const double h0 = fma(qa, Fi00, qb * Fi01), h1 = fma(qa, Fi01, qb * Fi11);
            const double ra = cls == 0 ? G0 : (cls == 1 ? -fa : (cls == 2 ? 0.0 : -h0));
            const double rb = cls == 0 ? G1 : (cls == 1 ? -fb : (cls == 2 ? 0.0 : -h1));
#pragma unroll
            for (int t = 0; t < 3; ++t) out[G_M + MLD * j + tq + 4 * t] = fma(ra, b0[t], fma(rb, b1[t], c0[t]));
        }
        // in-band marker (pad column 12 of row 0 of the chunk's slot 0): 1 + the slot of a failing step in this chunk, 0 if none
        if (lane == 0) pb[G_M + 12] = (double)(sfail + 1);
        __builtin_amdgcn_s_waitcnt(0xc07f);                     // lgkmcnt(0): the records are read, the publish buffer is written
        lds_store_flag(&flags[PF_BDONE], q + 1);
        lds_store_flag(&flags[PF_BREADY], (q + 1) | (last ? SH_FIN : 0));
        SH_TL(44, q, NCHK, true);
        if (last) { SH_MARK_MAX(20); return; }
    }
    // the chain diverged in a chunk it never finished counting: nothing more to build (PF_CREADY carried FIN with have <= q)
    lds_store_flag(&flags[PF_BREADY], (lds_load_flag(&flags[PF_BREADY]) & SH_CNT) | SH_FIN);
}

__device__ __forceinline__ void sh_publisher(const ShArgs &a, double *sm, const int gidx)
{
    const int lane = threadIdx.x % DDP_WAVE;
    const int N = a.N;
    int *flags = (int *)(sm + P_FLAGS);
    const double *pub = sm + P_PUB;
    const int NCHK = (N - 1) / CH + 1;
    double *grec = a.rec + (size_t)gidx * NCHK * GCHUNK;
    int *prog = &a.ctl->progress[16 * gidx];
    bool pending = false;                                       // chunk q - 1 is on its way, its progress word not yet written
    for (int q = 0;; ++q) {
        int r;
        for (;;) {
            r = lds_load_flag(&flags[PF_BREADY]);
            if ((r & SH_CNT) > q || (r & SH_FIN)) break;
            if (pending) {                                     // nothing to copy yet: finish the chunk before
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) store4_sc1(prog, q);
                pending = false;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        const int have = r & SH_CNT;
        if (have <= q) {                                       // finished without another chunk
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) store4_sc1(prog, have | SH_FIN);
            return;
        }
        const bool last = (r & SH_FIN) && have == q + 1;
        const double *pb = pub + (q % NPB) * GCHUNK;
        char *dst = (char *)(grec + (size_t)q * GCHUNK);
#pragma unroll
        for (int it = 0; it < (GPIECES + 63) / 64; ++it) {       // 22 stores (the last one 48 lanes)
            const int pc = 64 * it + lane, pcc = pc < GPIECES ? pc : GPIECES - 1;
            store16_sc1(dst + 16 * pcc, *(const d2 *)(pb + 2 * pcc));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        SH_TL(48, q, NCHK, gidx == 0);
        lds_store_flag(&flags[PF_PDONE], q + 1);                 // the publish buffer is free (its data is in the store queue)
        if (pending) {                                         // everything older than this chunk's 22 stores has left: chunk q - 1 is visible
            asm volatile("s_waitcnt vmcnt(22)" ::: "memory");
            if (lane == 0) store4_sc1(prog, q);
            pending = false;
        }
        if (last) {
            if (lane == 0) store4_sc1(&a.ctl->gdiverge[gidx], lds_load_flag(&flags[PF_CDIV]));
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // the chunk has left (write-through) before the progress word says so
            if (lane == 0) store4_sc1(prog, (q + 1) | SH_FIN);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            SH_MARK_MAX(21);
            return;
        }
        pending = true;
    }
}

// ============================================================= the consumer ==================================================
// chunk kinds of the consumer ring
enum { KIND_NORMAL = 0, KIND_ZERO = 1 << 20 /* no data: everything is zero */ };

// A wait that times out (a protocol error, or a producer that was starved of its CU for seconds) does not end in silently wrong
// results: the tile hands ITS trajectories to the per-trajectory kernels that the dispatcher launches behind this one (fb_active, the
// mask those kernels run under — they rewrite every output of a flagged trajectory), and ctl->error counts the event.
__device__ __forceinline__ void sh_dma(const ShArgs &a, double *sm, const int gidx, const int nusers, const int4 item)
{
    const int lane = threadIdx.x % DDP_WAVE;
    int *flags = (int *)(sm + C_FLAGS);
    const int NCHK = (a.N - 1) / CH + 1;
    const double *grec = a.rec + (size_t)gidx * NCHK * GCHUNK;
    const int *prog = &a.ctl->progress[16 * gidx];
    int pubd = 0;                                               // progress word as last seen
    const unsigned long long t0 = wall_clock64();
    // The gradients [cx; cu] of the tile's trajectories come through THIS wave too (round 5).  The affine waves used to fetch their own
    // by direct-to-LDS loads and wait for them with s_waitcnt vmcnt — a counter their result stores share: with the acknowledgement of a
    // store taking ~10 us under the launch's 3 TB/s of writes, "at most 3 outstanding" made every chunk wait for the stores of two chunks
    // before, the tiles ran three chunks (the depth of the ring) behind the chain and ended ~15 us after it.  This wave issues loads only,
    // so its count is exact: piece P = 64 k + lane of an affine wave's 4 x 48 pieces (trajectory P / 48; L = P % 48 < 40: cx doubles
    // 2L, 2L + 1 of the chunk, else cu of step L - 40), image q % NEI of that wave, freed with the ring slot of the records.
    const int cnt = item.z, naff = (cnt + 3) / 4, cTop = (a.N - 1) / CH, st = (a.N - 1) % CH;
    const char *gsrc[NAFF][3];
    int pstep[3], gstep[3];
#pragma unroll
    for (int kk = 0; kk < 3; ++kk) {
        const int P = 64 * kk + lane, tr = P / 48, L = P % 48;
        pstep[kk] = L < 40 ? L / 5 : L - 40;
        gstep[kk] = L < 40 ? CH * n * 8 : CH * m * 8;               // bytes per chunk
#pragma unroll
        for (int aw = 0; aw < NAFF; ++aw) {
            const int tt = 4 * aw + tr, t2 = tt < cnt ? tt : cnt - 1;
            const int bb = a.perm[item.y + t2];
            gsrc[aw][kk] = L < 40 ? (const char *)(a.cx + ((size_t)bb * a.N + (size_t)CH * cTop) * n) + 16 * L
                                  : (const char *)(a.cu + ((size_t)bb * a.N + (size_t)CH * cTop) * m) + 16 * (L - 40);
        }
    }
    auto wait_older_than = [&](int newest) {                    // s_waitcnt vmcnt(22 + 3 naff): the immediate must be a constant
        switch (newest) {
        case 25: asm volatile("s_waitcnt vmcnt(25)" ::: "memory"); break; case 28: asm volatile("s_waitcnt vmcnt(28)" ::: "memory"); break;
        case 31: asm volatile("s_waitcnt vmcnt(31)" ::: "memory"); break; case 34: asm volatile("s_waitcnt vmcnt(34)" ::: "memory"); break;
        case 37: asm volatile("s_waitcnt vmcnt(37)" ::: "memory"); break; case 40: asm volatile("s_waitcnt vmcnt(40)" ::: "memory"); break;
        case 43: asm volatile("s_waitcnt vmcnt(43)" ::: "memory"); break; case 46: asm volatile("s_waitcnt vmcnt(46)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    };
    static_assert(NAFF <= 8 && 22 + 3 * NAFF <= 63, "vmcnt is a 6-bit counter");
    int qwait = 0;
    auto give_up = [&]() {
        for (int t = lane; t < item.z; t += DDP_WAVE) a.fb_active[a.perm[item.y + t]] = 1;
        if (lane == 0) {
            atomicAdd(&a.ctl->error, 1); atomicAdd(&a.ctl->errors_total, 1);
            const int slot = atomicAdd(&a.ctl->ndiag, 1);
            if (slot < SH_NDIAG) {
                unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
                int *d = a.ctl->diag[slot];
                d[0] = (int)blockIdx.x; d[1] = gidx; d[2] = qwait; d[3] = pubd; d[4] = (int)((wall_clock64() - t0) / 100000ull);
                d[5] = (int)(xcc & 15u); d[6] = a.ctl->G; d[7] = a.ctl->launches;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // direct-to-LDS loads of earlier chunks may still be in flight
        lds_store_flag(&flags[CF_SREADY], SH_ABORT);
    };
    for (int q = 0; q < NCHK; ++q) {
        qwait = q;
        if (__builtin_expect(a.test_abort && q == 1, 0)) { give_up(); return; }       // DDP_TEST_SH_ABORT: every tile gives up at its second chunk
        // (1) the chunk must have been published (or the group has ended before it)
        while ((pubd & SH_CNT) <= q && !(pubd & SH_FIN)) {
            pubd = __builtin_amdgcn_readfirstlane(load4_sc1(prog));          // drains this wave's loads as well
            lds_store_flag(&flags[CF_SREADY], q);                             // chunks < q have landed
            if ((pubd & SH_CNT) <= q && !(pubd & SH_FIN)) {
                if (timed_out(t0)) { give_up(); return; }
                __builtin_amdgcn_s_sleep(20);
            }
        }
        SH_TL(52, q, NCHK, item.y == 0 && gidx == 0);
        // (2) ring slot q % NSB held chunk q - NSB: every user must be past it
        if (q >= NSB) {
            for (;;) {
                int lo = 1 << 28;
                for (int u = 0; u < nusers; ++u) { const int d = lds_load_flag(&flags[CF_UDONE + u]); lo = d < lo ? d : lo; }
                if (lo >= q - NSB + 1) break;
                __builtin_amdgcn_s_sleep(2);
            }
        }
        if ((pubd & SH_CNT) > q) {
            lds_store_flag(&flags[CF_KIND + q % NSB], KIND_NORMAL);
            const char *src = (const char *)(grec + (size_t)q * GCHUNK);
            double *dst = sm + C_SBUF + (q % NSB) * GCHUNK;
#pragma unroll
            for (int it = 0; it < (GPIECES + 63) / 64; ++it) {
                const int pc = 64 * it + lane;
                if (pc < GPIECES) __builtin_amdgcn_global_load_lds((glb_void *)(src + 16 * pc), (lds_void *)(dst + 128 * it), 16, 0, 16 /* sc1 */);
            }
            const int top = q == 0 ? st : CH - 1;                   // (steps past N - 1 are not touched)
#pragma unroll
            for (int aw = 0; aw < NAFF; ++aw) {
                if (aw < naff) {
                    double *img = sm + C_WAVE + aw * C_WSZ + (q % NEI) * EIMG;
#pragma unroll
                    for (int kk = 0; kk < 3; ++kk)
                        if (pstep[kk] <= top) __builtin_amdgcn_global_load_lds((glb_void *)(gsrc[aw][kk] - (long)gstep[kk] * q), (lds_void *)(img + 128 * kk), 16, 0, 0);
                }
            }
            // everything older than this chunk's 22 + 3 naff loads has landed
            wait_older_than(22 + 3 * naff);
            lds_store_flag(&flags[CF_SREADY], q);
        } else {
            lds_store_flag(&flags[CF_KIND + q % NSB], KIND_ZERO);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            lds_store_flag(&flags[CF_SREADY], q + 1);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_store_flag(&flags[CF_SREADY], NCHK);
    SH_MARK_MAX(23);
}

// waits until chunk q of the ring is ready; false: the launch was aborted
__device__ __forceinline__ bool sh_wait_chunk(int *flags, int q, int &seen)
{
    while (seen <= q) {
        seen = lds_load_flag(&flags[CF_SREADY]);
        if (seen & SH_ABORT) return false;
        if (seen <= q) __builtin_amdgcn_s_sleep(2);
    }
    asm volatile("" ::: "memory");
    return true;
}

template <bool NTS>
__device__ __forceinline__ void sh_affine(const ShArgs &a, double *sm, const int gidx, const int aw, const int4 item)
{
    const int lane = threadIdx.x % DDP_WAVE, j = lane & 15, r = lane >> 4;
    int *flags = (int *)(sm + C_FLAGS);
    const int N = a.N, cTop = (N - 1) / CH, st = (N - 1) % CH, NCHK = cTop + 1;
    double *wv = sm + C_WAVE + aw * C_WSZ;
    double *eimg = wv, *zcell = wv + NEI * EIMG, *dcell = wv + NEI * EIMG + 4;
    if (lane < 4) zcell[lane] = 0.0;
    // my row's trajectory (rows past the tile repeat its last trajectory and store nothing)
    const int cnt = item.z, tloc = 4 * aw + r, tl = tloc < cnt ? tloc : cnt - 1;
    const bool rowlive = tloc < cnt;
    const int b = a.perm[item.y + tl];
    // result image (= the gradient image the DMA wave filled, overwritten in place): piece P = 64 k + lane of the wave's 4 x 48 pieces;
    // trajectory P / 48, piece L = P % 48: L < 40: Vx doubles 2L, 2L + 1 of the chunk (step L / 5), else k of step L - 40
    char *rdst[3]; bool plive[3]; int pstep[3];
#pragma unroll
    for (int kk = 0; kk < 3; ++kk) {
        const int P = 64 * kk + lane, tr = P / 48, L = P % 48;
        const int tt = 4 * aw + tr, t2 = tt < cnt ? tt : cnt - 1;
        const int bb = a.perm[item.y + t2];
        plive[kk] = tt < cnt;
        pstep[kk] = L < 40 ? L / 5 : L - 40;
        rdst[kk] = L < 40 ? (char *)(a.Vx + ((size_t)bb * N + (size_t)CH * cTop) * n) + 16 * L
                          : (char *)(a.k + ((size_t)bb * N + (size_t)CH * cTop) * m) + 16 * (L - 40);
    }
    char *const sinkp = (char *)a.sink + 16 * lane;
    const int gstep[3] = {(lane % 48) < 40 ? CH * n * 8 : CH * m * 8, ((64 + lane) % 48) < 40 ? CH * n * 8 : CH * m * 8,
                          ((128 + lane) % 48) < 40 ? CH * n * 8 : CH * m * 8};          // bytes per chunk
    // per-lane offsets inside an image: lane j of row r reads / writes element j of step s at eo + es * s
    const int eo = 96 * r + (j < n ? j : (j < p ? 80 + (j - n) : 0)), es = j < n ? n : (j < p ? m : 0);
    const double *ebase = j < p ? eimg + eo : zcell;            // (image 0; image i is i * EIMG further)
    const int eflip = j < p ? EIMG : 0;
    double *rbase0 = j < p ? eimg + eo : dcell + (lane & 3);          // results overwrite the gradients of their step (same image, same cell)
    const double maskx = j < n ? 1.0 : 0.0;
    const int mrow = G_M + MLD * j;
    double s = 0.0, acc1 = 0.0, acc2 = 0.0;
    int seen = 0, gdiv = 0;
    bool ok = true;
    for (int q = 0; q < NCHK; ++q) {
        const int top = q == 0 ? st : CH - 1;
#ifdef SH_PROF
        const unsigned long long ta1 = __builtin_readcyclecounter();
#endif
        // (the chunk's records AND this wave's gradient image come through the tile's DMA wave: one flag says both have landed)
        if (!sh_wait_chunk(flags, q, seen)) { ok = false; break; }
#ifdef SH_PROF
        const unsigned long long ta2 = __builtin_readcyclecounter();
        SH_TL(56, q, NCHK, item.y == 0 && aw == 0 && gidx == 0);
#endif
        const int kind = lds_load_flag(&flags[CF_KIND + q % NSB]);
        const double *sb = sm + C_SBUF + (q % NSB) * GCHUNK;
        const double *eb = ebase + (q % NEI) * eflip;
        double *rbase = rbase0 + (q % NEI) * eflip;
        const double *res = eimg + (q % NEI) * EIMG;
        if (kind == KIND_NORMAL) {
            for (int slot = top; slot >= 0; --slot) {
                const double *mr = sb + GREC * slot + mrow;
                const d2 m0 = *(const d2 *)(mr), m1 = *(const d2 *)(mr + 2), m2 = *(const d2 *)(mr + 4), m3 = *(const d2 *)(mr + 6),
                         m4 = *(const d2 *)(mr + 8), m5 = *(const d2 *)(mr + 10);
                const double e = eb[es * slot];
                double y0 = 0.0, y1 = 0.0;
                fmac_bcast<0, 0xf, true>(y0, s, m0.x); fmac_bcast<1>(y1, s, m0.y);
                fmac_bcast<2>(y0, s, m1.x); fmac_bcast<3>(y1, s, m1.y);
                fmac_bcast<4>(y0, s, m2.x); fmac_bcast<5>(y1, s, m2.y);
                fmac_bcast<6>(y0, s, m3.x); fmac_bcast<7>(y1, s, m3.y);
                fmac_bcast<8>(y0, s, m4.x); fmac_bcast<9>(y1, s, m4.y);
                fmac_bcast<10>(y0, e, m5.x); fmac_bcast<11>(y1, e, m5.y);
                const double y = y0 + y1;
                fmac_bcast<10, 0xf, true>(acc1, y, y);          // lane 12: k0 Qu0, lane 14: k0 (Quu k)0
                fmac_bcast<11>(acc2, y, y);                     // lane 13: k1 Qu1, lane 15: k1 (Quu k)1
                s = fma(e, maskx, y);                           // lanes 0-9: Vx_i, 10-11: k_i
                rbase[es * slot] = s;
            }
        } else {
            for (int slot = top; slot >= 0; --slot) rbase[es * slot] = 0.0;
        }
        // the steps at and below a failing step are zero (:226-229); their records carry zero maps, so nothing entered dV
        if (kind == KIND_NORMAL) {
            const int mk = (int)sb[G_M + 12];                    // 1 + slot of the failing step of this chunk (0: none)
            if (__builtin_expect(mk != 0, 0)) {
                gdiv = CH * (cTop - q) + mk;
                for (int slot = top; slot >= 0; --slot) if (slot < mk) rbase[es * slot] = 0.0;
            }
        }
        // results of the chunk: 3 x 64 pieces, whole 640 / 128 byte bursts per trajectory
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        d2 rv[3];
#pragma unroll
        for (int kk = 0; kk < 3; ++kk) rv[kk] = *(const d2 *)(res + 128 * kk + 2 * lane);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int kk = 0; kk < 3; ++kk) store16_res<NTS>((plive[kk] && pstep[kk] <= top) ? rdst[kk] - (long)gstep[kk] * q : sinkp, rv[kk]);
        lds_store_flag(&flags[CF_UDONE + aw], q + 1);
        SH_TL(60, q, NCHK, item.y == 0 && aw == 0 && gidx == 0);
#ifdef SH_PROF
        if (item.y == 0 && aw == 0 && lane == 0 && q > 0) {       // first tile of group 0: where an affine wave's chunk period goes
            const unsigned long long ta3 = __builtin_readcyclecounter();
            a.ctl->prof[33] += ta2 - ta1; a.ctl->prof[34] += ta3 - ta2; a.ctl->prof[35] += 1;
            if (ta2 - ta1 < 200) a.ctl->prof[36] += 1;              // chunks that were already there when the wave asked for them
        }
#endif
    }
    if (!ok) { lds_store_flag(&flags[CF_UDONE + aw], 1 << 28); return; }
    // dV (:68) and diverge of my row's trajectory
    const double d1 = __shfl(acc1, 16 * r + 12) + __shfl(acc2, 16 * r + 13);
    const double d2v = __shfl(acc1, 16 * r + 14) + __shfl(acc2, 16 * r + 15);
    if (rowlive && j == 0) { a.dV[2 * (size_t)b] = d1; a.dV[2 * (size_t)b + 1] = 0.5 * d2v; a.diverge[b] = gdiv; }
    SH_MARK_MAX(24);
}

template <bool NTS>
__device__ __forceinline__ void sh_writer(const ShArgs &a, double *sm, const int gidx, const int ww, const int4 item)
{
    const int lane = threadIdx.x % DDP_WAVE;
    int *flags = (int *)(sm + C_FLAGS);
    const int N = a.N, cTop = (N - 1) / CH, st = (N - 1) % CH, NCHK = cTop + 1;
    const int cnt = item.z;
    constexpr size_t nn = (size_t)n * n, nm = (size_t)n * m, mm = (size_t)m * m;
    // instruction k < 6: Vxx pieces 64k + lane; 6: K pieces 0..63; 7: lanes 0-15 Vxx 384.., 16-31 K 64.., 32-47 Quu 0..15
    int loff[8], goff[8], gslot[8];                             // LDS offset (doubles) in a chunk, byte offset in the array's chunk, slot
#pragma unroll
    for (int kk = 0; kk < 6; ++kk) { const int pc = 64 * kk + lane; gslot[kk] = pc / 50; loff[kk] = GREC * (pc / 50) + 2 * (pc % 50); goff[kk] = 16 * pc; }
    { const int pc = lane; gslot[6] = pc / 10; loff[6] = GREC * (pc / 10) + G_K + 2 * (pc % 10); goff[6] = 16 * pc; }
    const int part = lane >> 4, l16 = lane & 15;
    {
        if (part == 0) { const int pc = 384 + l16; gslot[7] = pc / 50; loff[7] = GREC * (pc / 50) + 2 * (pc % 50); goff[7] = 16 * pc; }
        else if (part == 1) { const int pc = 64 + l16; gslot[7] = pc / 10; loff[7] = GREC * (pc / 10) + G_K + 2 * (pc % 10); goff[7] = 16 * pc; }
        else { const int pc = l16; gslot[7] = pc / 2; loff[7] = GREC * (pc / 2) + G_QUU + 2 * (pc % 2); goff[7] = 16 * pc; }
    }
    int seen = 0;
    for (int q = 0; q < NCHK; ++q) {
        const int top = q == 0 ? st : CH - 1;
        if (!sh_wait_chunk(flags, q, seen)) { lds_store_flag(&flags[CF_UDONE + NAFF + ww], 1 << 28); return; }
        const int kind = lds_load_flag(&flags[CF_KIND + q % NSB]);
        const double *sb = sm + C_SBUF + (q % NSB) * GCHUNK;
        d2 v[8];
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) v[kk] = kind == KIND_NORMAL ? *(const d2 *)(sb + loff[kk]) : d2{0.0, 0.0};
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        lds_store_flag(&flags[CF_UDONE + NAFF + ww], q + 1);    // the ring slot is free as far as this wave goes
        const size_t c8 = (size_t)CH * (cTop - q);
        for (int t = ww; t < cnt; t += NWR) {
            const int b = __builtin_amdgcn_readfirstlane(a.perm[item.y + t]);
            char *pV = (char *)(a.Vxx + ((size_t)b * N + c8) * nn), *pK = (char *)(a.K + ((size_t)b * N + c8) * nm),
                 *pQ = (char *)(a.Quu + ((size_t)b * N + c8) * mm);
#pragma unroll
            for (int kk = 0; kk < 6; ++kk) if (gslot[kk] <= top) store16_res<NTS>(pV + goff[kk], v[kk]);
            if (gslot[6] <= top) store16_res<NTS>(pK + goff[6], v[6]);
            char *p7 = part == 0 ? pV : (part == 1 ? pK : pQ);
            if (part < 3 && gslot[7] <= top) store16_res<NTS>(p7 + goff[7], v[7]);
        }
    }
}

template <bool REG2, bool NTS>
__global__ __launch_bounds__(SH_THREADS) void sh_back_kernel(ShArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    int *role_s = (int *)(sm + SH_LDS_DOUBLES - 2);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / DDP_WAVE);
    if (threadIdx.x == 0) *role_s = atomicAdd(&a.ctl->ticket, 1);
    {   // flags of both roles start at zero
        int *pf = (int *)(sm + P_FLAGS), *cf = (int *)(sm + C_FLAGS);
        if (threadIdx.x < 32) { pf[threadIdx.x] = 0; cf[threadIdx.x] = 0; }
    }
    __syncthreads();
    const int role = *role_s;
    const int G = a.ctl->G, W = a.ctl->W;
#ifdef SH_PROF
    if (threadIdx.x == 0) atomicMin(&a.ctl->prof[19], wall_clock64());
    struct Mark { ShCtl *c; bool on; __device__ ~Mark() { if (on) atomicMax(&c->prof[18], wall_clock64()); } } mark_{a.ctl, threadIdx.x % DDP_WAVE == 0};
#endif
    if (role < G) {
        if (wave > 2) return;
        const double lam = a.ctl->glam[role];
        if (wave == 0) sh_chain<REG2>(a, sm, role, lam);
        else if (wave == 1) sh_builder<REG2>(a, sm, lam);
        else sh_publisher(a, sm, role);
        return;
    }
    const int it = role - G;
    if (it >= W) return;
    const int4 item = a.items[it];
    const int naff = (item.z + 3) / 4;
#ifdef SH_PROF
    // per-tile end times (the record stream of group 15 is free in a profile run with one group): [4 it + {0 DMA, 1 affine 0, 2 last writer, 3 xcc}]
    struct TileMark { int *p; bool on; __device__ ~TileMark() { if (on) *p = (int)(wall_clock64() & 0x7fffffffull); } };
    int *tm = (int *)(a.rec + (size_t)15 * ((a.N - 1) / CH + 1) * GCHUNK) + 4 * it;
    const bool lane0 = threadIdx.x % DDP_WAVE == 0;
    TileMark tmk{tm + (wave == 0 ? 0 : (wave == 1 ? 1 : 2)), lane0 && (wave <= 1 || wave == NAFF + 1)};
    if (threadIdx.x == 0) { unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); tm[3] = (int)xcc; }
#endif
    if (wave == 0) sh_dma(a, sm, item.x, NAFF + NWR, item);
    else if (wave <= NAFF) {
        if (wave - 1 < naff) sh_affine<NTS>(a, sm, item.x, wave - 1, item);
        else { int *cf = (int *)(sm + C_FLAGS); if (threadIdx.x % DDP_WAVE == 0) lds_store_flag(&cf[CF_UDONE + wave - 1], 1 << 28); }
    } else sh_writer<NTS>(a, sm, item.x, wave - 1 - NAFF, item);
}

}   // namespace

#ifdef SH_PROF
static const int *g_sh_dbg = nullptr; static int g_sh_dbg_w = 0;
#endif

// The most consumer tiles sh_group_kernel can make of a batch of B trajectories on ncu compute units, over every number of groups
// G = 1 .. SH_GMAX and every grouped count start <= B: it chooses R = ceil(start / (slots TMAX)) rounds of slots = max(ncu - G, 8)
// tiles and raises the tile size until the tiles fit R * slots (or the tile size reaches TMAX, where the count is <= start / TMAX + G
// <= R * slots + G).  R * slots + G is largest at start = B; G is scanned.  (The round-4 bound assumed G = 1, start = B: with 16 groups
// of 500 on 256 CUs the device made 480 tiles for a 272-entry items[] and a 282-work-group grid.)
extern "C" int ddp_sh_max_tiles(int B, int ncu)
{
    int best = 0;
    for (int G = 1; G <= SH_GMAX; ++G) {
        const int slots = ncu - G > 8 ? ncu - G : 8;
        const int R = (B + slots * TMAX - 1) / (slots * TMAX);
        const int w = R * slots + G;
        best = w > best ? w : best;
    }
    const int floor_ = (B + TMAX - 1) / TMAX + SH_GMAX;          // what the device's clamp needs to terminate at T = TMAX
    return best > floor_ ? best : floor_;
}

// Shared-LTI backward pass.  Returns 1 when the shape is not handled here, 0 when launched (the caller then runs the per-trajectory
// kernels with *fb_active as their activity mask), < 0 on error.
int ddp_launch_back_pass_sh(ddp_handle h, const ddp_bp_desc *d, const double *cx, const double *cu,
                            const double *cxx, const double *cxu, const double *cuu, const double *fx,
                            const double *fu, const double *lambda, const int32_t *active, double *K,
                            double *k, double *Quu, double *Vx, double *Vxx, double *dV, int32_t *diverge,
                            const int32_t **fb_active)
{
    if (d->has_lims || d->m != 2 || d->n != 10 || d->fx_batched || d->cost_batched || d->fx_tv || d->cost_tv) return 1;
    if (d->N < 2 * CH || !h->sink) return 1;
    if ((((uintptr_t)cx | (uintptr_t)cu | (uintptr_t)K | (uintptr_t)k | (uintptr_t)Quu | (uintptr_t)Vx | (uintptr_t)Vxx | (uintptr_t)cxx | (uintptr_t)cuu) & 15) != 0) return 1;
    const int B = d->B, N = d->N;
    if (!h->ncu) { hipDeviceProp_t pr; DDP_HIP(hipGetDeviceProperties(&pr, h->device)); h->ncu = pr.multiProcessorCount; }
    const int Wmax = ddp_sh_max_tiles(B, h->ncu);
    const int NCHK = (N - 1) / CH + 1;
    // scratch of the handle: control block | items | perm | fb_active | record streams
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t o_items = al(sizeof(ShCtl)), o_perm = o_items + al(sizeof(int4) * (size_t)Wmax), o_fb = o_perm + al(sizeof(int) * (size_t)B),
                 o_rec = o_fb + al(sizeof(int32_t) * (size_t)B), total = o_rec + al(sizeof(double) * (size_t)SH_GMAX * NCHK * GCHUNK);
    if (h->sh_bytes < total) {
        if (h->sh) {
            DDP_HIP(hipStreamSynchronize(h->stream));
            ShCtl c; DDP_HIP(hipMemcpy(&c, h->sh, sizeof c, hipMemcpyDeviceToHost)); h->sh_timeouts += c.errors_total;
            DDP_HIP(hipFree(h->sh)); h->sh = nullptr; h->sh_bytes = 0;
        }
        DDP_HIP(hipMalloc(&h->sh, total));
        h->sh_bytes = total;
        DDP_HIP(hipMemsetAsync(h->sh, 0, sizeof(ShCtl), h->stream));
    }
    char *base = (char *)h->sh;
    ShArgs a;
    a.N = N; a.B = B; a.ncu = h->ncu; a.regType = d->regType; a.wmax = Wmax;
    a.test_abort = ddp_env(h, ENV_TEST_SH_ABORT) ? 1 : 0;
    a.cx = cx; a.cu = cu; a.cxx = cxx; a.cxu = cxu; a.cuu = cuu; a.fx = fx; a.fu = fu; a.lambda = lambda; a.active = active;
    a.K = K; a.k = k; a.Quu = Quu; a.Vx = Vx; a.Vxx = Vxx; a.dV = dV; a.diverge = diverge;
    a.ctl = (ShCtl *)base; a.items = (int4 *)(base + o_items); a.perm = (int *)(base + o_perm); a.fb_active = (int32_t *)(base + o_fb);
    a.rec = (double *)(base + o_rec);
    a.sink = (double *)h->sink;
    if (a.B <= 8 * 1024) hipLaunchKernelGGL(sh_group_kernel<true>, dim3(1), dim3(1024), 0, h->stream, a);
    else hipLaunchKernelGGL(sh_group_kernel<false>, dim3(1), dim3(1024), 0, h->stream, a);
    if (!h->sh_attr) {
        DDP_HIP(hipFuncSetAttribute((const void *)sh_back_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SH_LDS_BYTES));
        DDP_HIP(hipFuncSetAttribute((const void *)sh_back_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SH_LDS_BYTES));
        DDP_HIP(hipFuncSetAttribute((const void *)sh_back_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SH_LDS_BYTES));
        DDP_HIP(hipFuncSetAttribute((const void *)sh_back_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SH_LDS_BYTES));
        h->sh_attr = true;
    }
    const dim3 grid(SH_GMAX + Wmax), block(SH_THREADS);
    // result stores: non-temporal while the batch is small, plain from DDP_SH_NT_MAX_B trajectories on (default 3 072; store16_res)
    const char *nte = ddp_env(h, ENV_SH_NT_MAX_B);
    const bool nts = B < (nte ? atoi(nte) : 3072);
    if (d->regType == 2) { if (nts) hipLaunchKernelGGL((sh_back_kernel<true, true>), grid, block, SH_LDS_BYTES, h->stream, a); else hipLaunchKernelGGL((sh_back_kernel<true, false>), grid, block, SH_LDS_BYTES, h->stream, a); }
    else { if (nts) hipLaunchKernelGGL((sh_back_kernel<false, true>), grid, block, SH_LDS_BYTES, h->stream, a); else hipLaunchKernelGGL((sh_back_kernel<false, false>), grid, block, SH_LDS_BYTES, h->stream, a); }
    DDP_HIP(hipGetLastError());
    *fb_active = a.fb_active;
#ifdef SH_PROF
    g_sh_dbg = (const int *)(a.rec + (size_t)15 * NCHK * GCHUNK); g_sh_dbg_w = Wmax;
#endif
    return 0;
}

#ifdef SH_PROF
extern "C" int ddp_sh_prof_tiles(ddp_handle h, int *out, int cap)
{
    DDP_DEVICE(h);
    if (!g_sh_dbg) return -1;
    DDP_HIP(hipStreamSynchronize(h->stream));
    ShCtl c; DDP_HIP(hipMemcpy(&c, h->sh, sizeof c, hipMemcpyDeviceToHost));
    const int W = c.W < cap ? c.W : cap;
    DDP_HIP(hipMemcpy(out, g_sh_dbg, sizeof(int) * 4 * (size_t)W, hipMemcpyDeviceToHost));
    return W;
}
#endif
// -DSH_PROF builds (profiles/sh_phase_profile.py): the phase sums of the last launch; zeros in the production library
extern "C" int ddp_sh_prof(ddp_handle h, unsigned long long *out32)
{
    DDP_DEVICE(h);
    if (!h->sh) return -1;
    DDP_HIP(hipStreamSynchronize(h->stream));
    ShCtl c;
    DDP_HIP(hipMemcpy(&c, h->sh, sizeof c, hipMemcpyDeviceToHost));
    for (int e = 0; e < 64; ++e) out32[e] = c.prof[e];
    return 0;
}

// the tiles that gave up (ddp_sh_timeouts > 0): up to 8 records of 8 ints — {work-group, group, chunk it waited for, progress word it
// last saw (chunks published | 1 << 30 finished), milliseconds waited, XCD, groups of the launch, launch number} — followed by the 16
// progress words as they stand now; returns the number of records (0 in a healthy run).  out: 8 * 8 + 16 ints.
extern "C" int ddp_sh_timeout_info(ddp_handle h, int *out, int cap)
{
    DDP_DEVICE(h);
    if (!out || cap < SH_NDIAG * 8 + SH_GMAX) { ddp_set_error("ddp_sh_timeout_info: out needs %d ints", SH_NDIAG * 8 + SH_GMAX); return -2; }
    for (int e = 0; e < SH_NDIAG * 8 + SH_GMAX; ++e) out[e] = 0;
    if (!h->sh) return 0;
    DDP_HIP(hipStreamSynchronize(h->stream));
    ShCtl c;
    DDP_HIP(hipMemcpy(&c, h->sh, sizeof c, hipMemcpyDeviceToHost));
    const int nd = c.ndiag < SH_NDIAG ? c.ndiag : SH_NDIAG;
    for (int r = 0; r < nd; ++r) for (int e = 0; e < 8; ++e) out[8 * r + e] = c.diag[r][e];
    for (int g = 0; g < SH_GMAX; ++g) out[SH_NDIAG * 8 + g] = c.progress[16 * g];
    return nd;
}

extern "C" int ddp_sh_timeouts(ddp_handle h)
{
    DDP_DEVICE(h);
    if (!h->sh) return h->sh_timeouts;
    DDP_HIP(hipStreamSynchronize(h->stream));
    ShCtl c;
    DDP_HIP(hipMemcpy(&c, h->sh, sizeof c, hipMemcpyDeviceToHost));
    return h->sh_timeouts + c.errors_total;
}
