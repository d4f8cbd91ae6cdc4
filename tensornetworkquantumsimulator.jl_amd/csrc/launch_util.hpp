// launch_util.hpp -- host-side helpers shared by the kernel translation units.
#pragma once
#include <hip/hip_runtime_api.h>
#include <cstdlib>
#include <map>
#include <mutex>
#include <utility>

namespace tnqs {

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device, per-function attribute: remember it per (device, function) behind a lock
// (handles on different devices, used from different threads, each get the attribute set once on THEIR device).
inline void set_max_dynamic_lds(const void* func, size_t bytes) {
    static std::mutex mu;
    static std::map<std::pair<int, const void*>, size_t> done;
    int dev = 0; (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(mu);
    auto key = std::make_pair(dev, func);
    auto it = done.find(key);
    if (it != done.end() && it->second >= bytes) return;
    (void)hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    done[key] = bytes;
}

// (The f32 plane kernels form complex products with Gauss' three multiplications, mfma_common.hpp CAcc32; the K = N = 128 epilogue keeps four: its registers do not hold three accumulators per block.)

// f32 products of the chi = 32 plane kernels on the bf16 matrix cores (three-way exact split, six products: kernels_x3.hip); TNQS_NO_BF16X3=1 keeps them on
// v_mfma_f32_32x32x2_f32.
inline bool mfma_use_x3() { static const bool v = [] { const char* e = std::getenv("TNQS_NO_BF16X3"); return !(e && e[0] == '1'); }(); return v; }

}  // namespace tnqs
