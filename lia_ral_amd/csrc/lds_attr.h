// lds_attr.h -- the ONE place that raises a kernel's dynamic-LDS limit (hipFuncAttributeMaxDynamicSharedMemorySize).
//
// The attribute belongs to a (device, kernel) pair: a launch with more than 64 KB of dynamic LDS fails on every device
// where it was not raised.  The C ABI promises one context per GPU driven from any host thread (include/gmmiv.h, "thread
// safety"; the reference gives every worker thread its own servers, AccumulateTVStat.cpp:392-393, 422-423), so the "already
// done" state is kept per kernel AND per device, the fast path is one atomic load, the slow path (first launch of a
// kernel on a device, or a larger request) is serialised so that a smaller request never lowers a limit just raised.  `Kernel` is a template VALUE parameter -- two instantiations of one
// kernel template share a function type but never a `done` array.
//
// tests/test_cpu_plumbing.py greps csrc/ for hipFuncSetAttribute outside this header.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <mutex>

#define GMMIV_LDS_ATTR_MAX_DEV 64

template <auto Kernel> inline hipError_t gmmiv_lds_attr(size_t lds)
{
    static std::atomic<size_t> done[GMMIV_LDS_ATTR_MAX_DEV]; // zero-initialised; largest size set so far, per device
    int dev = -1;
    const hipError_t eg = hipGetDevice(&dev);
    if (eg != hipSuccess) return eg;
    const bool cached = dev >= 0 && dev < GMMIV_LDS_ATTR_MAX_DEV; // an ordinal beyond the table: set it every time
    if (cached && done[dev].load(std::memory_order_acquire) >= lds) return hipSuccess;
    static std::mutex slow; // the slow path only: a smaller request must never lower what another thread has just raised
    static size_t beyond = 0; // largest request seen for ordinals past the table (under `slow`)
    std::lock_guard<std::mutex> lock(slow);
    if (cached && done[dev].load(std::memory_order_acquire) >= lds) return hipSuccess;
    if (!cached) lds = beyond = lds > beyond ? lds : beyond;
    const hipError_t e = hipFuncSetAttribute((const void *)Kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    if (cached) done[dev].store(lds, std::memory_order_release);
    return hipSuccess;
}
