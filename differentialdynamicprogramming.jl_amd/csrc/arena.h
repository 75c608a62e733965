// arena.h — staging of host-pointer calls: carve the handle's scratch, queue uploads/downloads (not installed).
#pragma once
#include <utility>
#include <vector>
#include "ddp_internal.h"

struct Arena {
    ddp_handle h;
    char *base = nullptr;
    size_t off = 0, cap = 0;
    std::vector<std::pair<void *, std::pair<const void *, size_t>>> ups;     // dst, (src, bytes)
    std::vector<std::pair<void *, std::pair<void *, size_t>>> downs;          // host dst, (dev src, bytes)
    static size_t al(size_t b) { return (b + 255) & ~(size_t)255; }
    size_t need = 0;
    void want(size_t bytes) { need += al(bytes); }
    int commit()
    {
        void *p;
        int rc = ddp_scratch(h, need + 256, &p);
        if (rc) return rc;
        base = (char *)p; cap = need + 256; off = 0;
        return 0;
    }
    void *take(size_t bytes) { void *p = base + off; off += al(bytes); return p; }
    template <class T> const T *in(const T *host, size_t count)
    {
        if (!host) return nullptr;
        void *d = take(count * sizeof(T));
        ups.push_back({d, {host, count * sizeof(T)}});
        return (const T *)d;
    }
    template <class T> T *outp(T *host, size_t count)
    {
        if (!host) return nullptr;
        void *d = take(count * sizeof(T));
        downs.push_back({host, {d, count * sizeof(T)}});
        return (T *)d;
    }
    int upload()
    {
        for (auto &u : ups) DDP_HIP(hipMemcpyAsync(u.first, u.second.first, u.second.second, hipMemcpyHostToDevice, h->stream));
        return 0;
    }
    int download()
    {
        for (auto &d : downs) DDP_HIP(hipMemcpyAsync(d.first, d.second.first, d.second.second, hipMemcpyDeviceToHost, h->stream));
        DDP_HIP(hipStreamSynchronize(h->stream));
        return 0;
    }
};
