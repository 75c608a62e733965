// back_pass_mx_common.h — what the two fp64-MFMA-tile backward kernels (back_pass_mx.hip: one wave per trajectory;
// back_pass_mx2.hip: chain wave + write-back wave per trajectory) share: tile constants, the step record of the LDS groups,
// cross-lane helpers.  Included inside each file's anonymous namespace.
#pragma once

typedef double d4 __attribute__((ext_vector_type(4)));
typedef unsigned u2v __attribute__((ext_vector_type(2)));

struct BPXArgs {
    int N, B;
    int fx_batched, cost_batched;
    int n, m;                                       // run-time sizes (back_pass_mx_kernel<..., RT = true> only)
    const double *cx, *cu, *cxx, *cxu, *cuu, *fx, *fu, *lambda;
    const int32_t *active;
    double *K, *k, *Quu, *Vx, *Vxx, *dV;
    int32_t *diverge;
    const double *lims, *u;                         // control limits (back_pass_mxg_kernel<..., LIMS = true> only)
};

__device__ const double mx_zero[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
__device__ const double mx_one[2] = {1.0, 1.0};

constexpr int n = 10, m = 2, p = 12, VC = 12;      // VC: tile column that carries the vectors
constexpr int PD = 8;                               // prefetch distance (time steps) of the streamed operands
constexpr int TLD = 17;                             // leading dimension of the LDS transpose tile (odd: no bank conflicts)
constexpr int TZERO = TLD * 16;                     // a cell that stays 0.0
// LCH mode: the results of a group of PD time steps are collected in the LDS as one record per step, in the order they have in
// memory, and written back by 16-byte stores at the end of the group; the vector [cx;cu] of the group arrives by one
// direct-to-LDS load a group ahead.  A vector-memory instruction costs this lone wave ~50 issue cycles, an LDS access ~8.
constexpr int REC = 136;                            // doubles per step record: Vxx 100 | Vx 10 | K 20 | k 2 | Quu 4
constexpr int R_VX = 100, R_K = 110, R_KV = 130, R_QUU = 132;
constexpr int LOUT = REC * PD;                      // the records of a group; behind them the cells lanes without an output write to
constexpr int LDUMP = LOUT, LDUMP_SZ = 64 + 16 + REC * (PD - 1);
constexpr int EREC = 12;                            // [cx; cu] of a step
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;
typedef double d2 __attribute__((ext_vector_type(2)));

template <int L>
__device__ __forceinline__ double row_bcast(double x) { return __builtin_amdgcn_update_dpp(0.0, x, 0x150 + L, 0xf, 0xf, true); }

__device__ __forceinline__ double rcp_nr(double x)
{   // 1/x: hardware estimate + two Newton steps
    double y = __builtin_amdgcn_rcp(x);
    double e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    return y;
}

// z holds tile rows (10, 11, 10, 11) in the four 16-lane rows of the wave ("the u-row with the parity of my row"):
//   q0 <- row 10 everywhere, q1 <- row 11 everywhere   (one v_permlane16_swap per dword)
__device__ __forceinline__ void spread_pair(double z, double &q0, double &q1)
{
    const unsigned lo = (unsigned)__double2loint(z), hi = (unsigned)__double2hiint(z);
    const u2v e = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);      // .x = rows (0,0,2,2), .y = rows (1,1,3,3)
    const u2v f = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    q0 = __hiloint2double((int)f.x, (int)e.x);
    q1 = __hiloint2double((int)f.y, (int)e.y);
}

// acc += x[lane L of my 16-lane row] * y    (v_fmac_f64_dpp: the broadcast costs nothing); RM: the 16-lane rows that take part.
// The hazard recogniser does not look into inline asm, so the CALLER keeps two rules: (1) acc and x are never the
// in-flight result of an MFMA (pass such a value through a real VALU instruction first: `+ 0.0`); (2) x was not written by
// the VALU within the last two instructions — FENCE = true puts the two wait states in front when that cannot be ruled out;
// (3) an MFMA must not read acc right behind the asm (it does not know the asm wrote it): the statements are volatile, so
// they keep their place in the source order — S is written before the stores and the hand-off, W is never written here.
// (Measured: volatile is also 2 % faster than letting the scheduler move them.)
template <int L, int RM = 0xf, bool FENCE = false>
__device__ __forceinline__ void fmac_bcast(double &acc, double x, double y)
{
    if (FENCE) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:%4 bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(y), "n"(L), "n"(RM));
    else asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:%4 bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(y), "n"(L), "n"(RM));
}

// Stores of a lane subset without the compiler's save-exec / branch / restore sequence around each of them (the step is
// issue-bound, scalar instructions count).  The wave runs with all 64 lanes enabled everywhere else in the kernel.
__device__ __forceinline__ void store_masked(char *ptr, double a, unsigned long long lanes)
{
    asm volatile("s_mov_b64 exec, %2\n\tglobal_store_dwordx2 %0, %1, off\n\ts_mov_b64 exec, -1" ::"v"(ptr), "v"(a), "s"(lanes) : "memory");
}
__device__ __forceinline__ void store2_masked(char *ptr, double a, double b, unsigned long long lanes)
{
    asm volatile("s_mov_b64 exec, %3\n\tglobal_store_dwordx2 %0, %1, off\n\tglobal_store_dwordx2 %0, %2, off offset:32\n\ts_mov_b64 exec, -1"
                 ::"v"(ptr), "v"(a), "v"(b), "s"(lanes) : "memory");
}

typedef const __attribute__((address_space(1))) double *gdp;     // explicit global loads: a FLAT load forces s_waitcnt vmcnt(0)

template <int I> struct IC { static constexpr int value = I; };
template <int I, int E, class Fn>
__device__ __forceinline__ void static_for(Fn &&f)
{
    if constexpr (I < E) { f(IC<I>{}); static_for<I + 1, E>(f); }
}

struct Stream {          // a per-lane operand that moves by `stride` bytes per time step (0: time-invariant)
    const char *base;
    unsigned stride;
    const char *cur;     // cursor of the prefetcher
    __device__ __forceinline__ double at(int t) const { return *(gdp)(base + (size_t)stride * (unsigned)t); }
    __device__ __forceinline__ void seek(int t) { cur = base + (size_t)stride * (unsigned)t; }
    __device__ __forceinline__ double next() const { return *(gdp)cur; }
    __device__ __forceinline__ void back() { cur -= stride; }
};
