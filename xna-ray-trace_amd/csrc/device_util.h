// device_util.h — small device helpers shared by kernels.hip and packet.hip (gfx950, wave64).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/xrt.h"
#include "kernels.h"
#include "traverse.h"

namespace xrt {

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }
__device__ __forceinline__ int lanes_below(unsigned long long m) {
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}

// OR of v over the 64 lanes (DPP row shifts + the two row broadcasts, result read from lane 63): wave-uniform
__device__ __forceinline__ int wave_or(int v) {
    v |= __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v |= __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v |= __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v |= __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);   // row_shr:8
    v |= __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
    v |= __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3
    return __builtin_amdgcn_readlane(v, 63);
}

// A per-lane variable that lives in LDS instead of a register, [word][lane] (stride 64 words: one bank per lane, conflict free):
// what a kernel keeps for a whole query but touches rarely (k_intersect's scene cursor, k_packet's scene-level best answer).
template <class T>
struct LdsField {
    unsigned *p;
    __device__ __forceinline__ operator T() const { T v; __builtin_memcpy(&v, p, 4); return v; }
    __device__ __forceinline__ LdsField &operator=(T v) { __builtin_memcpy(p, &v, 4); return *this; }
    __device__ __forceinline__ LdsField &operator=(const LdsField &o) { *p = *o.p; return *this; }
    __device__ __forceinline__ T operator++(int) { T v = *this; *this = v + 1; return v; }
    __device__ __forceinline__ T operator--() { T v = (T)(*this) - 1; *this = v; return v; }
    __device__ __forceinline__ LdsField &operator&=(T m) { *this = (T)(*this) & m; return *this; }
};
struct LdsRay {
    unsigned *p;   // 9 words, stride 64 (the parallel-axis bits are recomputed from the direction)
    __device__ __forceinline__ operator RayPre() const {
        RayPre r;
        r.o = mk(i2f((int)p[0]), i2f((int)p[64]), i2f((int)p[128]));
        r.d = mk(i2f((int)p[192]), i2f((int)p[256]), i2f((int)p[320]));
        r.inv = mk(i2f((int)p[384]), i2f((int)p[448]), i2f((int)p[512]));
        r.par = (fabsf(r.d.x) < 1e-06f ? 1 : 0) | (fabsf(r.d.y) < 1e-06f ? 2 : 0) | (fabsf(r.d.z) < 1e-06f ? 4 : 0);   // as make_ray
        return r;
    }
    __device__ __forceinline__ LdsRay &operator=(const RayPre &r) {
        p[0] = (unsigned)f2i(r.o.x); p[64] = (unsigned)f2i(r.o.y); p[128] = (unsigned)f2i(r.o.z);
        p[192] = (unsigned)f2i(r.d.x); p[256] = (unsigned)f2i(r.d.y); p[320] = (unsigned)f2i(r.d.z);
        p[384] = (unsigned)f2i(r.inv.x); p[448] = (unsigned)f2i(r.inv.y); p[512] = (unsigned)f2i(r.inv.z);
        return *this;
    }
};
__device__ __forceinline__ void ray_axis(const LdsRay &w, int k, float &o, float &d, float &inv) {
    o = i2f((int)w.p[64 * k]); d = i2f((int)w.p[64 * (3 + k)]); inv = i2f((int)w.p[64 * (6 + k)]);
}
// Launch timing without events.  An event on a kernel's dispatch packet costs that launch and its successor ~5 us each on this
// stack (a 0.16 ms frame of ten kernels ran 0.135 ms without them), so the traversal kernels of a single-chunk frame time
// themselves on the 100 MHz device clock: wave 0 stamps the launch's start and its number of waves, every wave its own end;
// k_compose's frame epilogue folds a row into (start, latest end) for the host.  Row layout: kernels.h STAMP_*.
__device__ __forceinline__ void stamp_begin(unsigned long long *row) {
    if (row && blockIdx.x == 0 && threadIdx.x == 0) { row[0] = wall_clock64(); row[1] = gridDim.x * (blockDim.x >> 6); }
}
__device__ __forceinline__ void stamp_end(unsigned long long *row) {   // every wave, once, on its way out
    const unsigned slot = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row && lane_id() == 0 && slot < (unsigned)STAMP_SLOTS) row[STAMP_HEADER + slot] = wall_clock64();
}

struct alignas(16) Hit16 { int i0, i1, i2, i3; };

__device__ __forceinline__ void store_hit(xrt_hit *dst, const HitOut &h) {
    Hit16 *p = reinterpret_cast<Hit16 *>(dst);
    p[0] = Hit16{h.hit, h.object, h.mesh, h.tri};
    p[1] = Hit16{h.leaf, f2i(h.u), f2i(h.v), f2i(h.d)};
    p[2] = Hit16{f2i(h.wx), f2i(h.wy), f2i(h.wz), h.cost};   // (reserved word: scheduling feedback)
}
__device__ __forceinline__ void load_ray(const xrt_ray *src, v3 &o, v3 &d, int &im, int &it) {
    const f4 *p = reinterpret_cast<const f4 *>(src);
    f4 a = p[0], b = p[1];
    o = mk(a.x, a.y, a.z);
    d = mk(a.w, b.x, b.y);
    im = f2i(b.z);
    it = f2i(b.w);
}
__device__ __forceinline__ void store_ray(xrt_ray *dst, v3 o, v3 d, int im, int it) {
    f4 *p = reinterpret_cast<f4 *>(dst);
    p[0] = f4{o.x, o.y, o.z, d.x};
    p[1] = f4{d.y, d.z, i2f(im), i2f(it)};
}
constexpr int DEAD_RAY = -2;   // ignore_mesh marker of a path without a pixel (edge tiles)

// "Long ray first" scheduling.  One ray that skims a large mesh takes thousands of dependent steps, and a launch ends
// when its slowest ray does; started last, such a ray keeps a single wave alive long after the other 4095 have
// left.  Producers (k_raygen, k_shade) therefore estimate a ray's length inside the scene's root box, list the long
// ones and mark them in the ray record (bit 30 of ignore_tri set to the opposite of its sign bit); the traversal
// kernel takes the listed rays first and passes over them when it meets them again in the array.  Scheduling only:
// every ray is traced exactly once, by the same code.
constexpr int HEAVY_BIT = 0x40000000;
__device__ __forceinline__ bool heavy_marked(int ignoreTri) { return (((ignoreTri >> 30) ^ (ignoreTri >> 31)) & 1) != 0; }
}  // namespace xrt
