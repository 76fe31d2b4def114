// delta_group.h -- the XZ / 7-Zip Delta filter on the device, one stream per WORKGROUP.
//
// DeltaFilter.decode (reference Sources/Common/DeltaFilter.swift:11-33, called from XZBlock.swift:57-61 and 7zFolder.swift:
// 175-181):  out[i] = in[i] + delta[(distance + pos) % 256]  with a 256-byte ring written backwards, i.e.
// out[i] = in[i] + out[i - D] (bytes before the start are zero) with D = distance for 1..255 and D = 256 for distance 0 (the
// ring then returns the byte written 256 steps ago).  That is D independent running sums (one per residue class of i modulo
// D), and a running sum is a scan: the T threads of a group are split into D classes x C chunks; every thread adds up its
// chunk of its class, the chunk totals are scanned per class in LDS, and a second sweep writes the running sums.
#ifndef SWC_DELTA_GROUP_H
#define SWC_DELTA_GROUP_H

#include "swc_common.h"

namespace swc {
namespace delta {

template <int T>
struct Lds {
    uint8_t total[T];   // sum of the thread's chunk, then the sum of all earlier chunks of its class
};

// out[0 .. n) from in[0 .. n); in == out is allowed.  All T threads of the group call it.
template <int T>
SWC_D void delta_group(gcptr in, gptr out, uint64_t n, uint32_t distance, Lds<T>* l, int tid) {
    const uint32_t D = (distance & 255u) == 0 ? 256u : (distance & 255u);
    const uint32_t C = (uint32_t)T / D > 0 ? (uint32_t)T / D : 1u;      // chunks per class (T >= 256 => C >= 1)
    const uint32_t cls = (uint32_t)tid % D, chunk = (uint32_t)tid / D;
    const bool active = chunk < C && (uint32_t)tid < D * C;
    // class `cls` holds the elements cls, cls + D, ...: m of them, cut into C chunks of per elements
    const uint64_t m = n > cls ? (n - cls + D - 1) / D : 0;
    const uint64_t per = (m + C - 1) / C;
    const uint64_t lo = per * chunk < m ? per * chunk : m, hi = lo + per < m ? lo + per : m;
    uint32_t sum = 0;
    if (active) for (uint64_t k = lo; k < hi; k++) sum += in[cls + k * D];
    l->total[tid] = (uint8_t)sum;
    group_sync();
    uint32_t before = 0;
    if (active) for (uint32_t c = 0; c < chunk; c++) before += l->total[cls + c * D];
    group_sync();
    if (active) {
        uint32_t run = before;
        for (uint64_t k = lo; k < hi; k++) {
            run += in[cls + k * D];
            out[cls + k * D] = (uint8_t)run;
        }
    }
}

}  // namespace delta
}  // namespace swc
#endif
