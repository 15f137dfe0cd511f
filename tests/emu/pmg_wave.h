/* CPU stand-in of pybullet_multigoal_gym_amd/csrc/pmg_wave.h for the fiber emulator
 * (TEST INFRASTRUCTURE).  Same API, lane exchange through a shared buffer. */
#ifndef PMG_WAVE_H
#define PMG_WAVE_H
#include "hip_emu.h"

static inline int __float_as_int(float f) { int i; memcpy(&i, &f, 4); return i; }
static inline float __int_as_float(int i) { float f; memcpy(&f, &i, 4); return f; }
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __popc(unsigned x) { return __builtin_popcount(x); }
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
static inline float __builtin_amdgcn_fmed3f(float a, float b, float c) { return fmaxf(fminf(a, b), fminf(fmaxf(a, b), c)); }

namespace nt {   /* the product's non-temporal accesses: plain ones here */
static inline float4 load4(const float4* p) { return *p; }
static inline void store4(float4 r, float4* p) { *p = r; }
static inline float load(const float* p) { return *p; }
static inline void store(unsigned int v, unsigned int* p) { *p = v; }
}  // namespace nt

namespace wv {
constexpr int LANES = 64;
inline int lane() { return (int)threadIdx.x & 63; }
inline int lane_local() { return lane(); }
inline float* xbuf() { return emu_xf + ((int)threadIdx.x >> 6) * (64 * 16); } /* per-wave exchange area */
inline void lds_sync() { emu_barrier(); }
inline void set_priority(int) {}
template <class T>
inline T* as_lds(T* p) { return p; }
inline long long cycles() { static long long t = 0; return t += 64; }   /* (a counter: the diagnostics path stays deterministic) */

template <int N>
inline void exchange_put(const float* v)
{
    for (int k = 0; k < N; k++) xbuf()[64 * k + lane()] = v[k];
    emu_barrier();
}
inline void exchange_done() { emu_barrier(); }

template <int N>
inline void bcastn(const float* v, int src, float* out)
{
    exchange_put<N>(v);
    float t[N];
    for (int k = 0; k < N; k++) t[k] = xbuf()[64 * k + src];
    exchange_done();
    for (int k = 0; k < N; k++) out[k] = t[k];
}
inline float bcast(float v, int src)
{
    float o;
    bcastn<1>(&v, src, &o);
    return o;
}
inline int bcast_i(int v, int src) { return __float_as_int(bcast(__int_as_float(v), src)); }
template <int SRC>
inline float bcast_c(float v) { return bcast(v, SRC); }

template <class F>
inline float permute(float v, float fill, F srcfn)
{
    exchange_put<1>(&v);
    int s = srcfn(lane());
    float r = s >= 0 ? xbuf()[s] : fill;
    exchange_done();
    return r;
}
template <int N>
inline float row_shr(float v, float fill)
{
    return permute(v, fill, [](int l) { return (l & 15) >= N ? l - N : -1; });
}
template <int N>
inline float row_shl(float v, float fill)
{
    return permute(v, fill, [](int l) { return (l & 15) + N <= 15 ? l + N : -1; });
}
inline float row_sum(float v)
{
    exchange_put<1>(&v);
    float s = 0;
    int b = lane() & ~15;
    /* same association order as the DPP butterfly: ((a+b)+(c+d)) pairs, then halves */
    float q[4];
    for (int g = 0; g < 4; g++) {
        float a = xbuf()[b + 4 * g], bb = xbuf()[b + 4 * g + 1], c = xbuf()[b + 4 * g + 2], d = xbuf()[b + 4 * g + 3];
        q[g] = (a + bb) + (c + d);
    }
    s = (q[0] + q[1]) + (q[2] + q[3]);
    exchange_done();
    return s;
}
inline float half_sum(float v)
{
    exchange_put<1>(&v);
    int b = lane() & ~7;
    /* same association order as the DPP butterfly: pairs inside the quads, then the two quads */
    float s = ((xbuf()[b] + xbuf()[b + 1]) + (xbuf()[b + 2] + xbuf()[b + 3])) + ((xbuf()[b + 4] + xbuf()[b + 5]) + (xbuf()[b + 6] + xbuf()[b + 7]));
    exchange_done();
    return s;
}
inline float row_max(float v)
{
    exchange_put<1>(&v);
    int b = lane() & ~15;
    float s = xbuf()[b];
    for (int k = 1; k < 16; k++) s = fmaxf(s, xbuf()[b + k]);
    exchange_done();
    return s;
}
inline float sum_row0(float v) { return bcast(row_sum(v), 0); }
inline float max_row0(float v) { return bcast(row_max(v), 0); }
template <int NR>
inline float sum_rows(float v)
{
    v = row_sum(v);
    float s = bcast(v, 0);
    for (int r = 1; r < NR; r++) s += bcast(v, 16 * r);
    return s;
}
inline float sum_all(float v)
{
    v = row_sum(v);
    float t[4] = {bcast(v, 0), bcast(v, 16), bcast(v, 32), bcast(v, 48)};
    return (t[0] + t[1]) + (t[2] + t[3]);
}
/* the device's association order: (r1 + r0) and (r3 + r2) by row_bcast:15, then their sum by row_bcast:31 */
inline float sum_wave(float v)
{
    v = row_sum(v);
    float t[4] = {bcast(v, 0), bcast(v, 16), bcast(v, 32), bcast(v, 48)};
    return (t[3] + t[2]) + (t[1] + t[0]);
}
inline float max_all(float v)
{
    v = row_max(v);
    return fmaxf(fmaxf(bcast(v, 0), bcast(v, 16)), fmaxf(bcast(v, 32), bcast(v, 48)));
}
inline void opaque(int& i) { (void)i; }
inline void chain(int& off, float& done) { (void)off; (void)done; }
inline float sel_lane(float a, float b, int l, int lane_k) { return l == lane_k ? a : b; }
inline bool uniform_positive(float v) { return bcast(v, 0) > 0.f; } /* v_readfirstlane: lane 0 decides for the wave */
inline unsigned long long ballot(bool p)
{
    float f = p ? 1.f : 0.f;
    exchange_put<1>(&f);
    unsigned long long m = 0;
    for (int k = 0; k < 64; k++)
        if (xbuf()[k] != 0.f) m |= 1ull << k;
    exchange_done();
    return m;
}
template <int SRC8>
inline float half_bcast_c(float v)
{
    exchange_put<1>(&v);
    float r = xbuf()[(lane() & ~7) + SRC8];
    exchange_done();
    return r;
}
template <int SRC8>
inline void half_fma_bcast_c(float v, float c, float& acc) { acc = fmaf(c, half_bcast_c<SRC8>(v), acc); }
inline unsigned any_row_mask(bool p) { return (unsigned)(ballot(p) & 0xFFFFull); }
inline bool any_lane(bool p) { return ballot(p) != 0ull; }
template <int SRC>
inline void fma2_bcast_c(float v, float c1, float& acc1, float c2, float& acc2)
{
    const float b = bcast(v, SRC);
    acc1 = fmaf(c1, b, acc1);
    acc2 = fmaf(c2, b, acc2);
}
/* row-0 broadcasts of the device header: same values in the lanes that are allowed to look */
template <int SRC>
inline float bcast_r0_c(float v) { return bcast(v, SRC); }
inline float bcast_r0(float v, int src) { return bcast(v, src); }
template <int N>
inline void bcastn_r0(const float* v, int src, float* out) { bcastn<N>(v, src, out); }
template <int SRC>
inline void fma2_bcast_r0_c(float v, float c1, float& acc1, float c2, float& acc2) { fma2_bcast_c<SRC>(v, c1, acc1, c2, acc2); }
template <int SRC>
inline float dot6_bcast_r0_c(const float* v, const float* c)
{
    float acc = bcast(v[0], SRC) * c[0];
    for (int a = 1; a < 6; a++) acc = fmaf(bcast(v[a], SRC), c[a], acc);
    return acc;
}
template <int P>
inline void gj9_eliminate_r0_c(float* a, float f)
{
    for (int k = 1; k < 9; k++) { const int j = (P + k) % 9; a[j] = fmaf(bcast(a[j], P), f, a[j]); }
}
inline float rcp(float x) { return 1.f / x; }
template <int P>
inline void gj6_eliminate_r0_c(float* a, float& e, float f)
{
    for (int k = 1; k < 6; k++) { const int j = (P + k) % 6; a[j] = fmaf(bcast(a[j], P), f, a[j]); }
    e = fmaf(bcast(e, P), f, e);
}
inline float dot6_lanes_r0(const float* c, float v)
{
    float acc = bcast(v, 0) * c[0];
    for (int a = 1; a < 6; a++) acc = fmaf(bcast(v, a), c[a], acc);
    return acc;
}
inline float fsqrt(float x) { return sqrtf(x); }
inline float add_shr2_bank2(float y, float x)
{
    const float y2 = row_shr<2>(y, 0.f);
    const int rl = (int)threadIdx.x & 15;
    return (rl >= 8 && rl < 12) ? y2 + x : y;
}
}  // namespace wv

/* row-packed variant (four envs per wave, one per 16-lane row): see the product header.  The fiber scheduler
 * advances every live fiber by one barrier per pass, so rows whose control flow has diverged stay in lockstep
 * internally as long as an exchange only touches the caller's own row -- which is all wr does. */
namespace wr {
constexpr int LANES = 16;
inline int lane() { return (int)threadIdx.x & 15; }
inline int lane_local() { return lane(); }
inline int row() { return ((int)threadIdx.x & 63) >> 4; }
template <class T>
inline T* as_lds(T* p) { return p; }
inline int base() { return (int)threadIdx.x & 48; }
inline void lds_sync() { emu_barrier(); }
template <int N>
inline void bcastn(const float* v, int src, float* out)
{
    wv::exchange_put<N>(v);
    float t[N];
    for (int k = 0; k < N; k++) t[k] = wv::xbuf()[64 * k + base() + src];
    wv::exchange_done();
    for (int k = 0; k < N; k++) out[k] = t[k];
}
inline float bcast(float v, int src)
{
    float o;
    bcastn<1>(&v, src, &o);
    return o;
}
inline int bcast_i(int v, int src) { return __float_as_int(bcast(__int_as_float(v), src)); }
template <int SRC>
inline float bcast_c(float v) { return bcast(v, SRC); }
template <int N>
inline float row_shr(float v, float fill) { return wv::row_shr<N>(v, fill); }
template <int N>
inline float row_shl(float v, float fill) { return wv::row_shl<N>(v, fill); }
inline float row_sum(float v) { return wv::row_sum(v); }
inline float half_sum(float v) { return wv::half_sum(v); }
inline float row_max(float v) { return wv::row_max(v); }
inline float sum_row0(float v) { return wv::row_sum(v); }
inline float max_row0(float v) { return wv::row_max(v); }
template <int NR>
inline float sum_rows(float v) { return wv::row_sum(v); }
inline float sum_all(float v) { return wv::row_sum(v); }
inline float sum_wave(float v) { return wv::row_sum(v); }
inline float max_all(float v) { return wv::row_max(v); }
inline bool uniform_positive(float v) { return v > 0.f; }
inline unsigned long long ballot(bool p)
{
    float f = p ? 1.f : 0.f;
    wv::exchange_put<1>(&f);
    unsigned long long m = 0;
    for (int k = 0; k < 16; k++)
        if (wv::xbuf()[base() + k] != 0.f) m |= 1ull << k;
    wv::exchange_done();
    return m;
}
inline void opaque(int& i) { (void)i; }
inline void chain(int& off, float& done) { (void)off; (void)done; }
inline float sel_lane(float a, float b, int l, int lane_k) { return l == lane_k ? a : b; }
/* the device ORs the four rows (a wave-uniform mask that only decides which rows get a -- possibly empty -- visit);
 * rows of the emulator may have diverged, so each row answers for itself: same results, fewer empty visits */
template <int SRC8>
inline float half_bcast_c(float v) { return wv::half_bcast_c<SRC8>(v); }
template <int SRC8>
inline void half_fma_bcast_c(float v, float c, float& acc) { wv::half_fma_bcast_c<SRC8>(v, c, acc); }
inline unsigned any_row_mask(bool p) { return (unsigned)ballot(p); }
inline bool any_lane(bool p) { return ballot(p) != 0ull; }
template <int SRC>
inline void fma2_bcast_c(float v, float c1, float& acc1, float c2, float& acc2)
{
    const float b = bcast(v, SRC);
    acc1 = fmaf(c1, b, acc1);
    acc2 = fmaf(c2, b, acc2);
}
/* row-0 broadcasts of the device header: same values in the lanes that are allowed to look */
template <int SRC>
inline float bcast_r0_c(float v) { return bcast(v, SRC); }
inline float bcast_r0(float v, int src) { return bcast(v, src); }
template <int N>
inline void bcastn_r0(const float* v, int src, float* out) { bcastn<N>(v, src, out); }
template <int SRC>
inline void fma2_bcast_r0_c(float v, float c1, float& acc1, float c2, float& acc2) { fma2_bcast_c<SRC>(v, c1, acc1, c2, acc2); }
template <int SRC>
inline float dot6_bcast_r0_c(const float* v, const float* c)
{
    float acc = bcast(v[0], SRC) * c[0];
    for (int a = 1; a < 6; a++) acc = fmaf(bcast(v[a], SRC), c[a], acc);
    return acc;
}
template <int P>
inline void gj9_eliminate_r0_c(float* a, float f)
{
    for (int k = 1; k < 9; k++) { const int j = (P + k) % 9; a[j] = fmaf(bcast(a[j], P), f, a[j]); }
}
inline float rcp(float x) { return 1.f / x; }
template <int P>
inline void gj6_eliminate_r0_c(float* a, float& e, float f)
{
    for (int k = 1; k < 6; k++) { const int j = (P + k) % 6; a[j] = fmaf(bcast(a[j], P), f, a[j]); }
    e = fmaf(bcast(e, P), f, e);
}
inline float dot6_lanes_r0(const float* c, float v)
{
    float acc = bcast(v, 0) * c[0];
    for (int a = 1; a < 6; a++) acc = fmaf(bcast(v, a), c[a], acc);
    return acc;
}
inline float fsqrt(float x) { return sqrtf(x); }
inline float add_shr2_bank2(float y, float x)
{
    const float y2 = row_shr<2>(y, 0.f);
    const int rl = (int)threadIdx.x & 15;
    return (rl >= 8 && rl < 12) ? y2 + x : y;
}
}  // namespace wr
#endif
