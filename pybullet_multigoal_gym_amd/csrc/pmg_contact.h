/*
 * pmg_contact.h -- contacts of the batched env: box-box narrowphase (one pair
 * per lane), contact/friction constraint rows staged in LDS, and the
 * sequential-impulse visit of one row executed by the whole wavefront
 * (lane = DoF: lanes 0..8 robot joints, lanes 16+8b+c component c of block b).
 *
 * Restates, per env, what Bullet does inside stepSimulation for the pairs
 * that can touch in the reference's tasks: table x
 * block, block x block, finger x block, finger x table; rows and solver order
 * as in the oracle (oracle/pmg_oracle.c "collide", "row_setup", "substep").
 */
#ifndef PMG_CONTACT_H
#define PMG_CONTACT_H

#include "pmg_device.h"

namespace pmg {
#ifdef PMG_PROFILE
__device__ long long g_phase[16];
#define PMG_PH0() long long ph_t_ = wall_clock64()
#define PMG_PH(i) do { long long n_ = wall_clock64(); if (wv::lane() == 0 && blockIdx.x == 0) g_phase[i] += n_ - ph_t_; ph_t_ = wall_clock64(); } while (0)
#else
#define PMG_PH0() do { } while (0)
#define PMG_PH(i) do { } while (0)
#endif

constexpr float CONTACT_MARGIN = 0.002f;
constexpr float EDGE_FUDGE = 1.05f;
constexpr float BLOCK_MASS = (float)PMG_BLOCK_MASS;
constexpr float BLOCK_INERTIA = 0.0009f;      /* isotropic cube: PMG_BLOCK_INERTIA */
constexpr float BLOCK_HALF = 0.015f;
constexpr float FINGER_RADIUS = 0.0431f;      /* bounding sphere of the finger box + slack (oracle cull) */
constexpr int ROW_STRIDE = 40;                /* floats per constraint row in LDS (160 B: rows stay 16 B aligned) */
/* Row layouts.  SLOT (reach, block_stack): the Jacobian of at most two blocks sits in two slots and a
 * lane looks its slot up.  DIRECT (one free object: push / pick_and_place): lane l < 16 reads J[l]
 * and (M^-1 J^T)[l] straight from the row -- lanes 0..8 robot DoFs, lanes 9..14 the object's
 * [v, w] -- so a visit is two coalesced LDS reads, one DPP butterfly in row 0 and no lane lookup.
 *   SLOT  : [0..8] J robot | [9..14] J slot A | [15..20] J slot B | [21] id A | [22] id B | [23..31] M^-1J^T robot
 *   DIRECT: [0..8] J robot | [9..14] J object | [15] 0 | [16..24] M^-1J^T robot | [25..30] resp object | [31] 0
 *   both  : [32] rhs [33] 1/diag [34] applied impulse [35] mu (16 B aligned: one ds_read_b128)
 *           [36..38] direction (scratch of the row build) [39] has_robot                                  */
constexpr int ROW_SC = 32, ROW_RHS = 32, ROW_DINV = 33, ROW_APP = 34, ROW_MU = 35, ROW_DIR = 36, ROW_HASROB = 39;
constexpr int SLOT_IDA = 21, SLOT_IDB = 22;
template <int NB>
struct RowLayout {
    static constexpr bool direct = NB == 1;
    static constexpr int R_OFF = direct ? 16 : 23; /* start of M^-1 J^T (robot part) */
};

struct BoxPose { float c[3]; float R[9]; };

/* a contact point record: pa[3] pb[3] n[3] dist (n from B to A), 10 floats, written straight into LDS */
constexpr int CP = 10;

__device__ __forceinline__ int clip_poly(const float (*in)[2], int n, float (*out)[2], int axis, float sign, float lim)
{
    int m = 0;
    for (int i = 0; i < n; i++) {
        const float* a = in[i];
        const float* b = in[(i + 1) % n];
        float da = sign * a[axis] - lim, db = sign * b[axis] - lim;
        if (da <= 0.f) { out[m][0] = a[0]; out[m][1] = a[1]; m++; }
        if ((da < 0.f && db > 0.f) || (da > 0.f && db < 0.f)) {
            float t = da / (da - db);
            out[m][0] = a[0] + t * (b[0] - a[0]);
            out[m][1] = a[1] + t * (b[1] - a[1]);
            m++;
        }
    }
    return m;
}

/* SAT over the 15 axes + face clipping / edge-edge; <= 4 points; n points from B to A */
constexpr int BOX_WORK = 104; /* floats of LDS workspace per pair lane */
/* W: per-lane LDS workspace for the dynamically indexed arrays (private "scratch" memory would cost
 * an HBM-path round trip per access; LDS is ~10x closer) */
__device__ inline int box_box(const float* ca, const float* Ra, const float* ha, const float* cb, const float* Rb,
                              const float* hb, float margin, float* out, float* W)
{
    float (*A)[3] = (float (*)[3])(W + 0);
    float (*B)[3] = (float (*)[3])(W + 9);
    for (int i = 0; i < 3; i++)
        for (int a = 0; a < 3; a++) { A[i][a] = Ra[3 * a + i]; B[i][a] = Rb[3 * a + i]; }
    float d[3] = {cb[0] - ca[0], cb[1] - ca[1], cb[2] - ca[2]};
    float (*C)[3] = (float (*)[3])(W + 18);
    float (*Q)[3] = (float (*)[3])(W + 27);
    float dA[3], dB[3];
    for (int i = 0; i < 3; i++) {
        dA[i] = dot3(d, A[i]);
        dB[i] = dot3(d, B[i]);
        for (int j = 0; j < 3; j++) { C[i][j] = dot3(A[i], B[j]); Q[i][j] = fabsf(C[i][j]); }
    }
    float best = -1e30f;
    int code = -1;
    float nrm[3] = {0.f, 0.f, 0.f};
    for (int i = 0; i < 3; i++) {
        float s = fabsf(dA[i]) - (ha[i] + hb[0] * Q[i][0] + hb[1] * Q[i][1] + hb[2] * Q[i][2]);
        if (s > margin) return 0;
        if (s > best) { best = s; code = i; float sg = dA[i] < 0.f ? -1.f : 1.f; nrm[0] = sg * A[i][0]; nrm[1] = sg * A[i][1]; nrm[2] = sg * A[i][2]; }
    }
    for (int j = 0; j < 3; j++) {
        float s = fabsf(dB[j]) - (hb[j] + ha[0] * Q[0][j] + ha[1] * Q[1][j] + ha[2] * Q[2][j]);
        if (s > margin) return 0;
        if (s > best) { best = s; code = 3 + j; float sg = dB[j] < 0.f ? -1.f : 1.f; nrm[0] = sg * B[j][0]; nrm[1] = sg * B[j][1]; nrm[2] = sg * B[j][2]; }
    }
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
            float L[3];
            cross3(A[i], B[j], L);
            float len = sqrtf(dot3(L, L));
            if (len < 1e-6f) continue;
            float proj = dot3(d, L);
            float ra = ha[i1] * Q[i2][j] + ha[i2] * Q[i1][j];
            float rb = hb[j1] * Q[i][j2] + hb[j2] * Q[i][j1];
            float s = (fabsf(proj) - (ra + rb)) / len;
            if (s > margin) return 0;
            float pen = s < 0.f ? s * EDGE_FUDGE : s / EDGE_FUDGE;
            if (pen > best) {
                best = s; code = 6 + 3 * i + j;
                float sg = proj < 0.f ? -1.f : 1.f;
                nrm[0] = sg * L[0] / len; nrm[1] = sg * L[1] / len; nrm[2] = sg * L[2] / len;
            }
        }
    if (code < 0) return 0;
    if (code >= 6) {
        int i = (code - 6) / 3, j = (code - 6) % 3;
        float pa[3] = {ca[0], ca[1], ca[2]}, pb[3] = {cb[0], cb[1], cb[2]};
        for (int kx = 0; kx < 3; kx++) {
            if (kx != i) { float sg = dot3(nrm, A[kx]) > 0.f ? 1.f : -1.f; for (int a = 0; a < 3; a++) pa[a] += sg * ha[kx] * A[kx][a]; }
            if (kx != j) { float sg = dot3(nrm, B[kx]) > 0.f ? -1.f : 1.f; for (int a = 0; a < 3; a++) pb[a] += sg * hb[kx] * B[kx][a]; }
        }
        float r[3] = {pb[0] - pa[0], pb[1] - pa[1], pb[2] - pa[2]};
        float uaub = C[i][j], q1 = dot3(A[i], r), q2 = -dot3(B[j], r);
        float den = 1.f - uaub * uaub;
        float s = 0.f, t = 0.f;
        if (den > 1e-8f) { s = (q1 + uaub * q2) / den; t = (uaub * q1 + q2) / den; }
        for (int a = 0; a < 3; a++) {
            out[a] = pa[a] + s * A[i][a];
            out[3 + a] = pb[a] + t * B[j][a];
            out[6 + a] = -nrm[a];
        }
        out[9] = best;
        return 1;
    }
    bool refA = code < 3;
    const float *cr = refA ? ca : cb, *ci = refA ? cb : ca, *hr = refA ? ha : hb, *hi = refA ? hb : ha;
    float (*Rr)[3] = refA ? A : B;
    float (*Ri)[3] = refA ? B : A;
    int ax = refA ? code : code - 3;
    float nr[3];
    if (refA) { nr[0] = nrm[0]; nr[1] = nrm[1]; nr[2] = nrm[2]; } else { nr[0] = -nrm[0]; nr[1] = -nrm[1]; nr[2] = -nrm[2]; }
    int ia = 0;
    float bestd = -1.f;
    for (int kx = 0; kx < 3; kx++) {
        float dd = fabsf(dot3(nr, Ri[kx]));
        if (dd > bestd) { bestd = dd; ia = kx; }
    }
    float isg = dot3(nr, Ri[ia]) > 0.f ? -1.f : 1.f;
    int iu = (ia + 1) % 3, iv = (ia + 2) % 3, ru = (ax + 1) % 3, rv = (ax + 2) % 3;
    float fc[3];
    for (int a = 0; a < 3; a++) fc[a] = ci[a] + isg * hi[ia] * Ri[ia][a];
    float (*poly)[2] = (float (*)[2])(W + 36);
    float (*tmp)[2] = (float (*)[2])(W + 52);
    float vz[4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        float rel[3];
        const float su = (c == 0 || c == 3) ? 1.f : -1.f, sv = c < 2 ? 1.f : -1.f;
        for (int a = 0; a < 3; a++) rel[a] = fc[a] + su * hi[iu] * Ri[iu][a] + sv * hi[iv] * Ri[iv][a] - cr[a];
        poly[c][0] = dot3(rel, Rr[ru]);
        poly[c][1] = dot3(rel, Rr[rv]);
        vz[c] = dot3(rel, nr);
    }
    float e1u = poly[1][0] - poly[0][0], e1v = poly[1][1] - poly[0][1], e1z = vz[1] - vz[0];
    float e2u = poly[3][0] - poly[0][0], e2v = poly[3][1] - poly[0][1], e2z = vz[3] - vz[0];
    float det = e1u * e2v - e1v * e2u;
    float gu = 0.f, gv = 0.f;
    if (fabsf(det) > 1e-12f) { gu = (e1z * e2v - e2z * e1v) / det; gv = (e2z * e1u - e1z * e2u) / det; }
    float z0 = vz[0] - gu * poly[0][0] - gv * poly[0][1];
    int n = 4;
    n = clip_poly(poly, n, tmp, 0, 1.f, hr[ru]);
    n = clip_poly(tmp, n, poly, 0, -1.f, hr[ru]);
    n = clip_poly(poly, n, tmp, 1, 1.f, hr[rv]);
    n = clip_poly(tmp, n, poly, 1, -1.f, hr[rv]);
    float (*pts)[3] = (float (*)[3])(W + 68);
    float* sep = W + 92;
    int m = 0;
    for (int c = 0; c < n; c++) {
        float z = z0 + gu * poly[c][0] + gv * poly[c][1];
        float s = z - hr[ax];
        if (s > margin) continue;
        pts[m][0] = poly[c][0]; pts[m][1] = poly[c][1]; pts[m][2] = z;
        sep[m] = s;
        m++;
    }
    if (m == 0) return 0;
    int* sel = (int*)(W + 100);
    int ns = 0;
    if (m <= 4) {
        for (int c = 0; c < m; c++) sel[ns++] = c;
    } else {
        int i0 = 0;
        for (int c = 1; c < m; c++) if (sep[c] < sep[i0]) i0 = c;
        int i1 = -1; float bd = -1.f;
        for (int c = 0; c < m; c++) {
            if (c == i0) continue;
            float du = pts[c][0] - pts[i0][0], dv = pts[c][1] - pts[i0][1];
            float dd = du * du + dv * dv;
            if (dd > bd) { bd = dd; i1 = c; }
        }
        int i2 = -1, i3 = -1; float amax = 0.f, amin = 0.f;
        for (int c = 0; c < m; c++) {
            if (c == i0 || c == i1) continue;
            float ar = (pts[i1][0] - pts[i0][0]) * (pts[c][1] - pts[i0][1]) - (pts[i1][1] - pts[i0][1]) * (pts[c][0] - pts[i0][0]);
            if (ar > amax) { amax = ar; i2 = c; }
            if (ar < amin) { amin = ar; i3 = c; }
        }
        sel[ns++] = i0; sel[ns++] = i1;
        if (i2 >= 0) sel[ns++] = i2;
        if (i3 >= 0) sel[ns++] = i3;
    }
    for (int c = 0; c < ns; c++) {
        int s = sel[c];
        float pin[3], pref[3];
        for (int a = 0; a < 3; a++) {
            float base = cr[a] + pts[s][0] * Rr[ru][a] + pts[s][1] * Rr[rv][a];
            pin[a] = base + pts[s][2] * nr[a];
            pref[a] = base + hr[ax] * nr[a];
        }
        for (int a = 0; a < 3; a++) {
            if (refA) { out[CP * c + a] = pref[a]; out[CP * c + 3 + a] = pin[a]; out[CP * c + 6 + a] = -nr[a]; }
            else { out[CP * c + a] = pin[a]; out[CP * c + 3 + a] = pref[a]; out[CP * c + 6 + a] = nr[a]; }
        }
        out[CP * c + 9] = sep[s];
    }
    return ns;
}

/* 3-way select with a dynamic index but register operands */
__device__ __forceinline__ float sel3(int i, float a, float b, float c) { return i == 0 ? a : (i == 1 ? b : c); }

/* Register-only front end of box_box: the 15-axis SAT with compile-time indices and the common
 * face-contact case in which the incident face lies inside the reference face (block on table,
 * finger over table, ...) so that no clipping is needed.  Same arithmetic, ordering and output as
 * box_box(); everything else (edge-edge, partial overlap) falls through to the general routine. */
__device__ inline int box_box_fast(const float* ca, const float* Ra, const float* ha, const float* cb, const float* Rb,
                                   const float* hb, float margin, float* out, float* W)
{
    float A[3][3], B[3][3], HA[3], HB[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        HA[i] = ha[i]; HB[i] = hb[i];
#pragma unroll
        for (int a = 0; a < 3; a++) { A[i][a] = Ra[3 * a + i]; B[i][a] = Rb[3 * a + i]; }
    }
    float d[3] = {cb[0] - ca[0], cb[1] - ca[1], cb[2] - ca[2]};
    float C[3][3], Q[3][3], dA[3], dB[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        dA[i] = dot3(d, A[i]);
        dB[i] = dot3(d, B[i]);
#pragma unroll
        for (int j = 0; j < 3; j++) { C[i][j] = dot3(A[i], B[j]); Q[i][j] = fabsf(C[i][j]); }
    }
    float best = -1e30f;
    int code = -1;
    bool sep_axis = false;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        float s = fabsf(dA[i]) - (HA[i] + HB[0] * Q[i][0] + HB[1] * Q[i][1] + HB[2] * Q[i][2]);
        sep_axis = sep_axis || s > margin;
        if (s > best) { best = s; code = i; }
    }
#pragma unroll
    for (int j = 0; j < 3; j++) {
        float s = fabsf(dB[j]) - (HB[j] + HA[0] * Q[0][j] + HA[1] * Q[1][j] + HA[2] * Q[2][j]);
        sep_axis = sep_axis || s > margin;
        if (s > best) { best = s; code = 3 + j; }
    }
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
            float Lx[3];
            cross3(A[i], B[j], Lx);
            float len = sqrtf(dot3(Lx, Lx));
            float proj = dot3(d, Lx);
            float ra = HA[i1] * Q[i2][j] + HA[i2] * Q[i1][j];
            float rb = HB[j1] * Q[i][j2] + HB[j2] * Q[i][j1];
            bool valid = !(len < 1e-6f);
            float s = (fabsf(proj) - (ra + rb)) / len;
            sep_axis = sep_axis || (valid && s > margin);
            float pen = s < 0.f ? s * EDGE_FUDGE : s / EDGE_FUDGE;
            if (valid && pen > best) { best = s; code = 6 + 3 * i + j; }
        }
    if (sep_axis || code < 0) return 0;
    if (code >= 6) return box_box(ca, Ra, ha, cb, Rb, hb, margin, out, W);
    const bool refA = code < 3;
    const int ax = refA ? code : code - 3;
    /* reference / incident frames through selects (no dynamic register indexing) */
    float Rr[3][3], Ri[3][3], hr[3], hi[3], cr[3], ci[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        hr[k] = refA ? HA[k] : HB[k]; hi[k] = refA ? HB[k] : HA[k];
        cr[k] = refA ? ca[k] : cb[k]; ci[k] = refA ? cb[k] : ca[k];
#pragma unroll
        for (int a = 0; a < 3; a++) { Rr[k][a] = refA ? A[k][a] : B[k][a]; Ri[k][a] = refA ? B[k][a] : A[k][a]; }
    }
    float dax = sel3(ax, refA ? dA[0] : dB[0], refA ? dA[1] : dB[1], refA ? dA[2] : dB[2]);
    float sgn = dax < 0.f ? -1.f : 1.f; /* axis direction from A towards B */
    float rn[3], rU[3], rV[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        rn[a] = sel3(ax, Rr[0][a], Rr[1][a], Rr[2][a]);
        rU[a] = sel3(ax, Rr[1][a], Rr[2][a], Rr[0][a]);
        rV[a] = sel3(ax, Rr[2][a], Rr[0][a], Rr[1][a]);
    }
    float hrn = sel3(ax, hr[0], hr[1], hr[2]), hru = sel3(ax, hr[1], hr[2], hr[0]), hrv = sel3(ax, hr[2], hr[0], hr[1]);
    /* nr: outward reference-face normal pointing at the incident box */
    float nsg = refA ? sgn : -sgn;
    float nr[3] = {nsg * rn[0], nsg * rn[1], nsg * rn[2]};
    int ia = 0;
    float bestd = -1.f;
#pragma unroll
    for (int kx = 0; kx < 3; kx++) {
        float dd = fabsf(dot3(nr, Ri[kx]));
        if (dd > bestd) { bestd = dd; ia = kx; }
    }
    float in_[3], iU[3], iV[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        in_[a] = sel3(ia, Ri[0][a], Ri[1][a], Ri[2][a]);
        iU[a] = sel3(ia, Ri[1][a], Ri[2][a], Ri[0][a]);
        iV[a] = sel3(ia, Ri[2][a], Ri[0][a], Ri[1][a]);
    }
    float hin = sel3(ia, hi[0], hi[1], hi[2]), hiu = sel3(ia, hi[1], hi[2], hi[0]), hiv = sel3(ia, hi[2], hi[0], hi[1]);
    float isg = dot3(nr, in_) > 0.f ? -1.f : 1.f;
    float fc[3];
#pragma unroll
    for (int a = 0; a < 3; a++) fc[a] = ci[a] + isg * hin * in_[a];
    float pu[4], pv[4], vz[4];
    bool inside = true;
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const float su = (c == 0 || c == 3) ? 1.f : -1.f, sv = c < 2 ? 1.f : -1.f;
        float rel[3];
#pragma unroll
        for (int a = 0; a < 3; a++) rel[a] = fc[a] + su * hiu * iU[a] + sv * hiv * iV[a] - cr[a];
        pu[c] = dot3(rel, rU);
        pv[c] = dot3(rel, rV);
        vz[c] = dot3(rel, nr);
        inside = inside && fabsf(pu[c]) <= hru && fabsf(pv[c]) <= hrv;
    }
    if (!inside) return box_box(ca, Ra, ha, cb, Rb, hb, margin, out, W);
    float e1u = pu[1] - pu[0], e1v = pv[1] - pv[0], e1z = vz[1] - vz[0];
    float e2u = pu[3] - pu[0], e2v = pv[3] - pv[0], e2z = vz[3] - vz[0];
    float det = e1u * e2v - e1v * e2u;
    float gu = 0.f, gv = 0.f;
    if (fabsf(det) > 1e-12f) { gu = (e1z * e2v - e2z * e1v) / det; gv = (e2z * e1u - e1z * e2u) / det; }
    float z0 = vz[0] - gu * pu[0] - gv * pv[0];
    int ns = 0;
#pragma unroll
    for (int c = 0; c < 4; c++) {
        float z = z0 + gu * pu[c] + gv * pv[c];
        float sp = z - hrn;
        if (sp > margin) continue;
        float* o = out + CP * ns;
#pragma unroll
        for (int a = 0; a < 3; a++) {
            float base = cr[a] + pu[c] * rU[a] + pv[c] * rV[a];
            float pin = base + z * nr[a], pref = base + hrn * nr[a];
            o[a] = refA ? pref : pin;
            o[3 + a] = refA ? pin : pref;
            o[6 + a] = refA ? -nr[a] : nr[a];
        }
        o[9] = sp;
        ns++;
    }
    return ns;
}

/* ---------------------------------------------------------------- */
/* cylinder (A) x box (B): gripper-base x block and the slide puck x table / finger.  Finite
 * separating-axis search (3 box face normals, cylinder axis, 3 axis x edge, 1 closest feature) and
 * feature clipping to <= 4 points, as oracle/pmg_oracle.c cyl_box.  n from the box to the cylinder.
 * W: per-lane LDS workspace (>= 84 floats): pts[12][3] sep[12] q8[8][3] sv[8] sel[4].           */
__device__ __forceinline__ float box_proj(const float (*B)[3], const float* hb, const float* L)
{
    return hb[0] * fabsf(dot3(B[0], L)) + hb[1] * fabsf(dot3(B[1], L)) + hb[2] * fabsf(dot3(B[2], L));
}
__device__ __forceinline__ void closest_on_box(const float* cb, const float (*B)[3], const float* hb, const float* p, float* q)
{
    float d[3] = {p[0] - cb[0], p[1] - cb[1], p[2] - cb[2]};
    q[0] = cb[0]; q[1] = cb[1]; q[2] = cb[2];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float t = fminf(fmaxf(dot3(d, B[k]), -hb[k]), hb[k]);
        q[0] += t * B[k][0]; q[1] += t * B[k][1]; q[2] += t * B[k][2];
    }
}
__device__ inline int reduce4(const float (*pts)[3], const float* sep, int m, int* sel)
{
    if (m <= 4) { for (int c = 0; c < m; c++) sel[c] = c; return m; }
    int i0 = 0;
    for (int c = 1; c < m; c++) if (sep[c] < sep[i0]) i0 = c;
    int i1 = -1; float bd = -1.f;
    for (int c = 0; c < m; c++) {
        if (c == i0) continue;
        float d[3] = {pts[c][0] - pts[i0][0], pts[c][1] - pts[i0][1], pts[c][2] - pts[i0][2]};
        float dd = dot3(d, d);
        if (dd > bd) { bd = dd; i1 = c; }
    }
    int i2 = -1, i3 = -1; float amax = 0.f, amin = 0.f;
    float e[3] = {pts[i1][0] - pts[i0][0], pts[i1][1] - pts[i0][1], pts[i1][2] - pts[i0][2]};
    float ref[3] = {0.f, 0.f, 0.f};
    for (int c = 0; c < m; c++) {
        if (c == i0 || c == i1) continue;
        float f[3] = {pts[c][0] - pts[i0][0], pts[c][1] - pts[i0][1], pts[c][2] - pts[i0][2]}, x[3];
        cross3(e, f, x);
        if (dot3(ref, ref) == 0.f && dot3(x, x) > 0.f) { ref[0] = x[0]; ref[1] = x[1]; ref[2] = x[2]; }
        float ar = dot3(x, ref);
        if (ar > amax) { amax = ar; i2 = c; }
        if (ar < amin) { amin = ar; i3 = c; }
    }
    int ns = 0;
    sel[ns++] = i0; sel[ns++] = i1;
    if (i2 >= 0) sel[ns++] = i2;
    if (i3 >= 0) sel[ns++] = i3;
    return ns;
}
__device__ __forceinline__ int cyl_box(const float* cc, const float* Rc, float rad, float hl, const float* cb, const float* Rb,
                                           float hbx, float hby, float hbz, float margin, float* out, float* W)
{
    float a[3] = {Rc[2], Rc[5], Rc[8]}, u[3] = {Rc[0], Rc[3], Rc[6]}, v[3] = {Rc[1], Rc[4], Rc[7]};
    float B[3][3], hb[3] = {hbx, hby, hbz}; /* extents by value: the (cold) call never forces a caller array onto the stack */
#pragma unroll
    for (int k = 0; k < 3; k++)
#pragma unroll
        for (int x = 0; x < 3; x++) B[k][x] = Rb[3 * x + k];
    float d[3] = {cc[0] - cb[0], cc[1] - cb[1], cc[2] - cb[2]};
    float best = -1e30f, n[3] = {0.f, 0.f, 0.f};
    int btype = -1, bk = 0;
#pragma unroll
    for (int pass = 0; pass < 8; pass++) {
        float L[3];
        int type, k = 0;
        bool valid = true;
        if (pass < 3) { type = 0; k = pass; L[0] = B[pass][0]; L[1] = B[pass][1]; L[2] = B[pass][2]; }
        else if (pass == 3) { type = 1; L[0] = a[0]; L[1] = a[1]; L[2] = a[2]; }
        else if (pass < 7) {
            type = 2; k = pass - 4;
            cross3(a, B[pass - 4], L);
            float len = sqrtf(dot3(L, L));
            valid = !(len < 1e-6f);
            float il = valid ? 1.f / len : 0.f;
            L[0] *= il; L[1] *= il; L[2] *= il;
        } else {
            type = 3;
            float p0[3], s0[3];
            closest_on_box(cb, B, hb, cc, p0);
            float w[3] = {p0[0] - cc[0], p0[1] - cc[1], p0[2] - cc[2]};
            float t = fminf(fmaxf(dot3(w, a), -hl), hl);
            s0[0] = cc[0] + t * a[0]; s0[1] = cc[1] + t * a[1]; s0[2] = cc[2] + t * a[2];
            closest_on_box(cb, B, hb, s0, p0);
            L[0] = s0[0] - p0[0]; L[1] = s0[1] - p0[1]; L[2] = s0[2] - p0[2];
            float len = sqrtf(dot3(L, L));
            valid = !(len < 1e-9f);
            float il = valid ? 1.f / len : 0.f;
            L[0] *= il; L[1] *= il; L[2] *= il;
        }
        if (!valid) continue;
        float t = dot3(d, L), ca = dot3(a, L);
        float rc = hl * fabsf(ca) + rad * sqrtf(fmaxf(1.f - ca * ca, 0.f));
        float sp = fabsf(t) - (box_proj(B, hb, L) + rc);
        if (sp > margin) return 0;
        float pen = type >= 2 ? (sp < 0.f ? sp * EDGE_FUDGE : sp / EDGE_FUDGE) : sp;
        if (pen > best) {
            best = sp; btype = type; bk = k;
            float sg = t < 0.f ? -1.f : 1.f;
            n[0] = sg * L[0]; n[1] = sg * L[1]; n[2] = sg * L[2];
        }
    }
    if (btype < 0) return 0;
    float (*pts)[3] = (float (*)[3])(W + 0);
    float* sep = W + 36;
    int m = 0;
    float can = dot3(a, n);
    if (btype == 0) {
        float bp = box_proj(B, hb, n);
        float fp[3] = {cb[0] + bp * n[0], cb[1] + bp * n[1], cb[2] + bp * n[2]};
        const float* Bk = bk == 0 ? B[0] : (bk == 1 ? B[1] : B[2]);
        const float* B1 = bk == 0 ? B[1] : (bk == 1 ? B[2] : B[0]);
        const float* B2 = bk == 0 ? B[2] : (bk == 1 ? B[0] : B[1]);
        float hk = bk == 0 ? hb[0] : (bk == 1 ? hb[1] : hb[2]);
        float h1 = bk == 0 ? hb[1] : (bk == 1 ? hb[2] : hb[0]);
        float h2 = bk == 0 ? hb[2] : (bk == 1 ? hb[0] : hb[1]);
        if (fabsf(can) >= 0.7f) {
            float sg = can > 0.f ? -1.f : 1.f;
            float pc[3] = {cc[0] + sg * hl * a[0], cc[1] + sg * hl * a[1], cc[2] + sg * hl * a[2]};
            for (int c = 0; c < 4; c++) {
                float ku = c == 0 ? rad : (c == 1 ? -rad : 0.f), kv = c == 2 ? rad : (c == 3 ? -rad : 0.f);
                float p[3] = {pc[0] + ku * u[0] + kv * v[0], pc[1] + ku * u[1] + kv * v[1], pc[2] + ku * u[2] + kv * v[2]};
                float w[3] = {p[0] - cb[0], p[1] - cb[1], p[2] - cb[2]};
                if (fabsf(dot3(w, B1)) <= h1 && fabsf(dot3(w, B2)) <= h2) {
                    pts[m][0] = p[0]; pts[m][1] = p[1]; pts[m][2] = p[2];
                    float w2[3] = {p[0] - fp[0], p[1] - fp[1], p[2] - fp[2]};
                    sep[m] = dot3(w2, n); m++;
                }
            }
            float nb = dot3(n, Bk) > 0.f ? 1.f : -1.f;
            for (int c = 0; c < 4; c++) {
                float s1 = (c & 1) ? h1 : -h1, s2 = (c & 2) ? h2 : -h2;
                float q[3];
                for (int x = 0; x < 3; x++) q[x] = cb[x] + nb * hk * Bk[x] + s1 * B1[x] + s2 * B2[x];
                float w[3] = {q[0] - pc[0], q[1] - pc[1], q[2] - pc[2]};
                float wa = dot3(w, a);
                if (dot3(w, w) - wa * wa <= rad * rad) {
                    float t = -wa / can;
                    pts[m][0] = q[0] + t * n[0]; pts[m][1] = q[1] + t * n[1]; pts[m][2] = q[2] + t * n[2];
                    sep[m] = t; m++;
                }
            }
            if (m == 0) {
                float q[3];
                closest_on_box(cb, B, hb, pc, q);
                float w[3] = {q[0] - pc[0], q[1] - pc[1], q[2] - pc[2]};
                float wa = dot3(w, a);
                w[0] -= wa * a[0]; w[1] -= wa * a[1]; w[2] -= wa * a[2];
                float rho = sqrtf(dot3(w, w));
                float sc = rho > 1e-9f ? fminf(rho, rad) / rho : 0.f;
                float p[3] = {pc[0] + sc * w[0], pc[1] + sc * w[1], pc[2] + sc * w[2]};
                float w2[3] = {p[0] - fp[0], p[1] - fp[1], p[2] - fp[2]};
                pts[m][0] = p[0]; pts[m][1] = p[1]; pts[m][2] = p[2]; sep[m] = dot3(w2, n); m++;
            }
        } else {
            float md[3] = {n[0] - can * a[0], n[1] - can * a[1], n[2] - can * a[2]};
            float ml = 1.f / sqrtf(dot3(md, md));
            md[0] *= ml; md[1] *= ml; md[2] *= ml;
            for (int e2 = 0; e2 < 2; e2++) {
                float he = e2 == 0 ? hl : -hl;
                float p[3] = {cc[0] - rad * md[0] + he * a[0], cc[1] - rad * md[1] + he * a[1], cc[2] - rad * md[2] + he * a[2]};
                float w[3] = {p[0] - fp[0], p[1] - fp[1], p[2] - fp[2]};
                float sd = dot3(w, n);
                if (sd > margin) continue;
                float w2[3] = {p[0] - cb[0], p[1] - cb[1], p[2] - cb[2]};
                float c1 = dot3(w2, B1), c2 = dot3(w2, B2);
                float k1 = fminf(fmaxf(c1, -h1), h1), k2 = fminf(fmaxf(c2, -h2), h2);
                for (int x = 0; x < 3; x++) p[x] += (k1 - c1) * B1[x] + (k2 - c2) * B2[x];
                pts[m][0] = p[0]; pts[m][1] = p[1]; pts[m][2] = p[2]; sep[m] = sd; m++;
            }
        }
    } else if (btype == 1) {
        float pc[3] = {cc[0] - hl * n[0], cc[1] - hl * n[1], cc[2] - hl * n[2]};
        float (*q8)[3] = (float (*)[3])(W + 48);
        float* sv = W + 72;
        float smax = -1e30f;
        for (int c = 0; c < 8; c++) {
            for (int x = 0; x < 3; x++)
                q8[c][x] = cb[x] + ((c & 1) ? hb[0] : -hb[0]) * B[0][x] + ((c & 2) ? hb[1] : -hb[1]) * B[1][x] + ((c & 4) ? hb[2] : -hb[2]) * B[2][x];
            float w[3] = {q8[c][0] - cb[0], q8[c][1] - cb[1], q8[c][2] - cb[2]};
            sv[c] = dot3(w, n);
            smax = fmaxf(smax, sv[c]);
        }
        for (int c = 0; c < 8 && m < 8; c++) {
            if (sv[c] < smax - 1e-3f) continue;
            float w[3] = {q8[c][0] - pc[0], q8[c][1] - pc[1], q8[c][2] - pc[2]};
            float wa = dot3(w, n);
            if (dot3(w, w) - wa * wa > rad * rad) continue;
            pts[m][0] = q8[c][0] - wa * n[0]; pts[m][1] = q8[c][1] - wa * n[1]; pts[m][2] = q8[c][2] - wa * n[2];
            sep[m] = -wa; m++;
        }
        if (m == 0) {
            float q[3];
            closest_on_box(cb, B, hb, pc, q);
            float w[3] = {q[0] - pc[0], q[1] - pc[1], q[2] - pc[2]};
            float wa = dot3(w, n);
            pts[m][0] = q[0] - wa * n[0]; pts[m][1] = q[1] - wa * n[1]; pts[m][2] = q[2] - wa * n[2];
            sep[m] = -wa; m++;
        }
    } else {
        float p0[3], s0[3] = {cc[0], cc[1], cc[2]};
        closest_on_box(cb, B, hb, cc, p0);
        for (int it = 0; it < 4; it++) {
            float w[3] = {p0[0] - cc[0], p0[1] - cc[1], p0[2] - cc[2]};
            float t = fminf(fmaxf(dot3(w, a), -hl), hl);
            s0[0] = cc[0] + t * a[0]; s0[1] = cc[1] + t * a[1]; s0[2] = cc[2] + t * a[2];
            closest_on_box(cb, B, hb, s0, p0);
        }
        pts[0][0] = s0[0] - rad * n[0]; pts[0][1] = s0[1] - rad * n[1]; pts[0][2] = s0[2] - rad * n[2];
        sep[0] = best;
        m = 1;
    }
    {
        int k2 = 0;
        for (int c = 0; c < m; c++)
            if (sep[c] <= margin) {
                if (k2 != c) { pts[k2][0] = pts[c][0]; pts[k2][1] = pts[c][1]; pts[k2][2] = pts[c][2]; sep[k2] = sep[c]; }
                k2++;
            }
        m = k2;
        if (m == 0) return 0;
    }
    int* sel = (int*)(W + 80);
    int ns = reduce4((const float (*)[3])pts, sep, m, sel);
    for (int c = 0; c < ns; c++) {
        int i = sel[c];
        float* o = out + CP * c;
        for (int x = 0; x < 3; x++) { o[x] = pts[i][x]; o[3 + x] = pts[i][x] - sep[i] * n[x]; o[6 + x] = n[x]; }
        o[9] = sep[i];
    }
    return ns;
}

/* [BULLET-PRIOR] btPlaneSpace1 */
__device__ __forceinline__ void plane_space(const float* n, float* p, float* q)
{
    if (fabsf(n[2]) > 0.7071067811865475244f) {
        float a = n[1] * n[1] + n[2] * n[2];
        float k = 1.f / sqrtf(a);
        p[0] = 0.f; p[1] = -n[2] * k; p[2] = n[1] * k;
        q[0] = a * k; q[1] = -n[0] * p[2]; q[2] = n[0] * p[1];
    } else {
        float a = n[0] * n[0] + n[1] * n[1];
        float k = 1.f / sqrtf(a);
        p[0] = -n[1] * k; p[1] = n[0] * k; p[2] = 0.f;
        q[0] = -n[2] * p[1]; q[1] = n[2] * p[0]; q[2] = a * k;
    }
}

__device__ __forceinline__ void quat_to_R(const float* q, float* R)
{
    float x = q[0], y = q[1], z = q[2], w = q[3];
    float d = x * x + y * y + z * z + w * w;
    float s = 2.f / d;
    float xs = x * s, ys = y * s, zs = z * s;
    float wx = w * xs, wy = w * ys, wz = w * zs, xx = x * xs, xy = x * ys, xz = x * zs, yy = y * ys, yz = y * zs, zz = z * zs;
    R[0] = 1.f - (yy + zz); R[1] = xy - wz; R[2] = xz + wy;
    R[3] = xy + wz; R[4] = 1.f - (xx + zz); R[5] = yz - wx;
    R[6] = xz - wy; R[7] = yz + wx; R[8] = 1.f - (xx + yy);
}

/* body ids in contacts */
constexpr int BODY_STATIC = -1;
constexpr int BODY_FINGER1 = 5, BODY_FINGER2 = 6, BODY_GBASE = 7; /* blocks are 0..4; 5..7 ride on the robot */

/* per-env LDS of the contact path */
template <int NB, int MAXC>
struct ContactLds {
    static constexpr int NPAIR = NB + NB * (NB - 1) / 2 + 2 * (NB + 1) + NB;
    float blk[NB > 0 ? NB : 1][BLOCK_DIM];    /* pos3 quat4 vel3 omg3 (persistent over the substeps) */
    float blkR[NB > 0 ? NB : 1][9];
    float fing[2][12];                         /* finger box centre + rotation */
    float gbase[12];                           /* gripper-base cylinder centre + rotation (link 7) */
    float S[NJ][6];
    float qd[NJ];
    float minv[NJ][NJ];
    int pair_count[NPAIR];
    float con[MAXC][12];                       /* a b pa3 pb3 n3 dist -> [0]=a [1]=b [2..4]pa [5..7]pb [8..10]n [11]dist */
    float con_mu[MAXC];
    /* the narrowphase scratch (box_box workspace + staged points) is dead once the contacts are
     * compacted into `con`, and the rows are built after that: they share storage */
    union {
        struct {
            float work[NPAIR][BOX_WORK];       /* box_box workspace per pair lane */
            float stage[NPAIR][4][10];         /* pa pb n dist per staged point */
        };
        float rows[3 * MAXC][ROW_STRIDE];
    };
    int ncon;
};

template <int NB>
__device__ __forceinline__ void decode_pair(int i, int nb, int& a, int& b)
{
    /* order: object x table | block x block (b < c) | finger f: objects..., table | gripper base x blocks */
    if (i < nb) { a = i; b = BODY_STATIC; return; }
    i -= nb;
    int nbb = nb * (nb - 1) / 2;
    if (i < nbb) {
        int x = 0;
        for (int p = 0; p < nb; p++)
            for (int q2 = p + 1; q2 < nb; q2++) {
                if (x == i) { a = p; b = q2; return; }
                x++;
            }
    }
    i -= nbb;
    if (i < 2 * (nb + 1)) {
        int f = i / (nb + 1), r = i % (nb + 1);
        a = f == 0 ? BODY_FINGER1 : BODY_FINGER2;
        b = r < nb ? r : BODY_STATIC;
        return;
    }
    i -= 2 * (nb + 1);
    a = BODY_GBASE; /* gripper-base cylinder x block i */
    b = i;
}

/* lowest z of the oriented finger box (exact AABB extent): a separating-axis bound for finger x table */
__device__ __forceinline__ float finger_zmin(const float* c, const float* R)
{
    const float fh[3] = PMG_FINGER_HALF;
    return c[2] - (fabsf(R[6]) * fh[0] + fabsf(R[7]) * fh[1] + fabsf(R[8]) * fh[2]);
}

/* collision detection for every candidate pair of this env; fills L.con / L.ncon (uniform) */
/* compile-time properties of the free object(s): the 3 cm cubes (block.urdf) or the slide puck
 * (cylinder_bulk.urdf; half = r, r, h/2).  A template parameter rather than kernel arguments so the
 * box tasks keep their constant-folded extents and scalar-register budget. */
template <bool CYL>
struct ObjT {
    static constexpr bool cyl = CYL;
    static constexpr float mu = CYL ? (float)PMG_PUCK_FRICTION : (float)PMG_BLOCK_FRICTION;
    __host__ __device__ static constexpr float half(int a)
    {
        constexpr float ph[3] = PMG_PUCK_HALF;
        return CYL ? ph[a] : BLOCK_HALF;
    }
    __host__ __device__ static constexpr float inv_inertia(int a) /* principal, body frame */
    {
        constexpr double pi[3] = PMG_PUCK_INERTIA;
        return CYL ? (float)(1.0 / pi[a]) : 1.f / BLOCK_INERTIA;
    }
};

template <int NB, int MAXC, bool CYL>
__device__ __forceinline__ int collide(ContactLds<NB, MAXC>& L, int nb, const float* table_c, const float* table_h, float table_mu)
{
    using OB = ObjT<CYL>;
    PMG_PH0();
    int l = wv::lane();
    int npair = nb + nb * (nb - 1) / 2 + 2 * (nb + 1) + (OB::cyl ? 0 : nb);
    const float I3[9] = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f};
    const float fh[3] = PMG_FINGER_HALF;
    const float oh[3] = {OB::half(0), OB::half(1), OB::half(2)};
    for (int i = l; i < npair; i += 64) {
        int a, b;
        decode_pair<NB>(i, nb, a, b);
        float tc[3] = {table_c[0], table_c[1], table_c[2]}, th[3] = {table_h[0], table_h[1], table_h[2]};
        /* every pair funnels into ONE box_box_fast and ONE cyl_box call site, so the lanes of a wave run the
         * narrowphase together instead of once per pair type.  kind: -1 culled, 0 box x box, 1 cylinder(A) x box(B);
         * swapped: the cylinder is body B of the pair (finger x puck), roles exchanged afterwards */
        const float *ca, *Ra, *ha, *cb, *Rb, *hb;
        int kind = 0;
        bool swapped = false;
        float rad = oh[0], hl = oh[2];
        if (a == BODY_GBASE) {
            ca = L.gbase; Ra = L.gbase + 3; ha = oh; rad = (float)PMG_GBASE_RADIUS; hl = (float)PMG_GBASE_HALFLEN;
            cb = L.blk[b]; Rb = L.blkR[b]; hb = oh;
            /* cull in the cylinder frame: axial and radial slabs grown by the block's bounding sphere */
            constexpr float rb = 0.026f + CONTACT_MARGIN;
            constexpr float rlim = (float)PMG_GBASE_RADIUS + rb;
            float dd[3] = {cb[0] - ca[0], cb[1] - ca[1], cb[2] - ca[2]};
            float az = Ra[2] * dd[0] + Ra[5] * dd[1] + Ra[8] * dd[2];
            float r2 = dot3(dd, dd) - az * az;
            kind = (fabsf(az) <= (float)PMG_GBASE_HALFLEN + rb && r2 <= rlim * rlim) ? 1 : -1;
        } else if (a >= BODY_FINGER1) {
            ca = L.fing[a - BODY_FINGER1]; Ra = ca + 3; ha = fh;
            if (b == BODY_STATIC) {
                cb = tc; Rb = I3; hb = th;
                if (!(finger_zmin(ca, Ra) < tc[2] + th[2] + CONTACT_MARGIN)) kind = -1;
            } else {
                cb = L.blk[b]; Rb = L.blkR[b]; hb = oh;
                /* cull in the finger frame: object bounding sphere against the finger box grown by it (conservative
                 * -- an excluded pair has a finger face axis separating it by more than the margin) */
                constexpr float rb = (CYL ? 0.0317f : 0.026f) + CONTACT_MARGIN;
                float dd[3] = {cb[0] - ca[0], cb[1] - ca[1], cb[2] - ca[2]};
                float lx = Ra[0] * dd[0] + Ra[3] * dd[1] + Ra[6] * dd[2];
                float ly = Ra[1] * dd[0] + Ra[4] * dd[1] + Ra[7] * dd[2];
                float lz = Ra[2] * dd[0] + Ra[5] * dd[1] + Ra[8] * dd[2];
                if (!(fabsf(lx) <= fh[0] + rb && fabsf(ly) <= fh[1] + rb && fabsf(lz) <= fh[2] + rb)) kind = -1;
                else if (CYL) { /* the puck is the cylinder: run it as A, exchange roles afterwards */
                    kind = 1; swapped = true;
                    const float* t;
                    t = ca; ca = cb; cb = t;
                    t = Ra; Ra = Rb; Rb = t;
                    hb = fh;
                }
            }
        } else if (b == BODY_STATIC) {
            ca = L.blk[a]; Ra = L.blkR[a]; ha = oh;
            cb = tc; Rb = I3; hb = th;
            kind = CYL ? 1 : 0;
        } else {
            ca = L.blk[a]; Ra = L.blkR[a]; ha = oh;
            cb = L.blk[b]; Rb = L.blkR[b]; hb = oh;
            float dd[3] = {ca[0] - cb[0], ca[1] - cb[1], ca[2] - cb[2]};
            if (dot3(dd, dd) > 0.06f * 0.06f) kind = -1;
        }
        float* stage = &L.stage[i][0][0];
        int n = 0;
        if (kind == 0) n = box_box_fast(ca, Ra, ha, cb, Rb, hb, CONTACT_MARGIN, stage, L.work[i]);
        if (kind == 1) {
            n = cyl_box(ca, Ra, rad, hl, cb, Rb, hb[0], hb[1], hb[2], CONTACT_MARGIN, stage, L.work[i]);
            if (swapped)
                for (int c = 0; c < n; c++) {
                    float* o = stage + CP * c;
                    for (int x = 0; x < 3; x++) { float t = o[x]; o[x] = o[3 + x]; o[3 + x] = t; o[6 + x] = -o[6 + x]; }
                }
        }
        L.pair_count[i] = n;
    }
    wv::lds_sync();
    PMG_PH(0);
    int total = 0;
    for (int i = l; i < npair; i += 64) {
        int off = 0;
        for (int j = 0; j < i; j++) off += L.pair_count[j];
        int n = L.pair_count[i];
        int a, b;
        decode_pair<NB>(i, nb, a, b);
        float mua = a == BODY_GBASE ? 0.5f : (a >= BODY_FINGER1 ? (float)PMG_FINGER_FRICTION : OB::mu);
        float mu = mua * (b == BODY_STATIC ? table_mu : OB::mu);
        for (int c = 0; c < n && off + c < MAXC; c++) {
            float* o = L.con[off + c];
            const float* st = L.stage[i][c];
            o[0] = (float)a; o[1] = (float)b;
            for (int k = 0; k < 10; k++) o[2 + k] = st[k];
            L.con_mu[off + c] = mu;
        }
    }
    for (int j = 0; j < npair; j++) total += L.pair_count[j];
    if (total > MAXC) total = MAXC;
    wv::lds_sync();
    PMG_PH(1);
    return total;
}

/* LDS row of (contact c, direction t): t = 0 normal, 1..2 friction */
template <int NB, int MAXC>
__device__ __forceinline__ float* row_of(ContactLds<NB, MAXC>& L, int c, int t)
{
    return t == 0 ? L.rows[c] : L.rows[MAXC + 2 * c + (t - 1)];
}

/* Build the normal + 2 friction rows of every contact in four lane-parallel phases:
 *   R1 lane = contact:            friction directions, block-side Jacobians / 1/mass terms / velocities
 *   R2 lane = (row, DoF) item:    robot Jacobian entries J[r][d] = (w_d x p + v_d) . dir
 *   R3 lane = (row, DoF) item:    (M^-1 J^T)[r][i] = sum_j M^-1[i][j] J[r][j]
 *   R4 lane = contact:            1/diag, relative velocity, right-hand sides
 * ([BULLET-PRIOR] btMultiBodyConstraintSolver::setupMultiBodyContactConstraint) */
template <int NB, int MAXC, bool CYL>
__device__ __forceinline__ void build_contact_rows(ContactLds<NB, MAXC>& L, int nc)
{
    using OB = ObjT<CYL>;
    using LY = RowLayout<NB>;
    PMG_PH0();
    int l = wv::lane();
    /* R1 */
    for (int c = l; c < nc; c += 64) {
        const float* o = L.con[c];
        int a = (int)o[0], b = (int)o[1];
        float n[3] = {o[8], o[9], o[10]}, t1[3], t2[3];
        plane_space(n, t1, t2);
        for (int t = 0; t < 3; t++) {
            float* row = row_of(L, c, t);
            const float* dir = t == 0 ? n : (t == 1 ? t1 : t2);
            for (int k = 0; k < ROW_STRIDE; k++) row[k] = 0.f;
            row[ROW_DIR] = dir[0]; row[ROW_DIR + 1] = dir[1]; row[ROW_DIR + 2] = dir[2];
            float denom = 0.f, rel = 0.f;
            if (!LY::direct) { row[SLOT_IDA] = -1.f; row[SLOT_IDB] = -1.f; }
            for (int s2 = 0; s2 < 2; s2++) {
                int id = s2 == 0 ? a : b;
                if (id < 0 || id >= BODY_FINGER1) continue;
                float sg = s2 == 0 ? 1.f : -1.f;
                const float* pt = o + 2 + 3 * s2;
                const float* bl = L.blk[id];
                float r[3] = {pt[0] - bl[0], pt[1] - bl[1], pt[2] - bl[2]}, rxn[3];
                cross3(r, dir, rxn);
                float* J = row + 9 + (LY::direct ? 0 : 6 * s2);
                for (int k = 0; k < 3; k++) { J[k] = sg * dir[k]; J[3 + k] = sg * rxn[k]; }
                /* angular response R diag(1/I) R^T (r x dir): isotropic for the cubes, general for the puck */
                float da[3] = {J[3] * OB::inv_inertia(0), J[4] * OB::inv_inertia(0), J[5] * OB::inv_inertia(0)};
                if (OB::cyl) {
                    float la[3];
                    const float* Rm = L.blkR[id];
                    la[0] = (Rm[0] * J[3] + Rm[3] * J[4] + Rm[6] * J[5]) * OB::inv_inertia(0);
                    la[1] = (Rm[1] * J[3] + Rm[4] * J[4] + Rm[7] * J[5]) * OB::inv_inertia(1);
                    la[2] = (Rm[2] * J[3] + Rm[5] * J[4] + Rm[8] * J[5]) * OB::inv_inertia(2);
                    mat3v(Rm, la, da);
                }
                if (LY::direct) {
                    for (int k = 0; k < 3; k++) { row[25 + k] = J[k] / BLOCK_MASS; row[28 + k] = da[k]; }
                } else {
                    row[SLOT_IDA + s2] = (float)id;
                }
                denom += dot3(J, J) / BLOCK_MASS + dot3(J + 3, da);
                rel += dot3(J, bl + 7) + dot3(J + 3, bl + 10);
            }
            row[ROW_DINV] = denom;
            row[ROW_RHS] = rel;
            row[ROW_HASROB] = (a >= BODY_FINGER1 || b >= BODY_FINGER1) ? 1.f : 0.f;
        }
    }
    wv::lds_sync();
    PMG_PH(2);
    /* R2 */
    for (int item = l; item < 27 * nc; item += 64) {
        int c = item / 27, t = (item / 9) % 3, d = item % 9;
        const float* o = L.con[c];
        float* row = row_of(L, c, t);
        float acc = 0.f;
        for (int s2 = 0; s2 < 2; s2++) {
            int id = (int)o[s2];
            if (id < BODY_FINGER1) continue;
            int own = id == BODY_FINGER1 ? 7 : (id == BODY_FINGER2 ? 8 : -1); /* the gripper base rides on link 7 only */
            if (d >= 7 && d != own) continue;
            const float* S = L.S[d];
            const float* pt = o + 2 + 3 * s2;
            float v[3];
            cross3(S, pt, v);
            v[0] += S[3]; v[1] += S[4]; v[2] += S[5];
            acc += (s2 == 0 ? 1.f : -1.f) * (v[0] * row[ROW_DIR] + v[1] * row[ROW_DIR + 1] + v[2] * row[ROW_DIR + 2]);
        }
        row[d] = acc;
    }
    wv::lds_sync();
    PMG_PH(3);
    /* R3 */
    for (int item = l; item < 27 * nc; item += 64) {
        int c = item / 27, t = (item / 9) % 3, i = item % 9;
        float* row = row_of(L, c, t);
        if (row[ROW_HASROB] != 0.f) {
            float s2 = 0.f;
#pragma unroll
            for (int j = 0; j < NJ; j++) s2 += L.minv[i][j] * row[j];
            row[LY::R_OFF + i] = s2;
        }
    }
    wv::lds_sync();
    PMG_PH(4);
    /* R4 */
    for (int c = l; c < nc; c += 64) {
        float dist = L.con[c][11] + LINEAR_SLOP;
        for (int t = 0; t < 3; t++) {
            float* row = row_of(L, c, t);
            float denom = row[ROW_DINV], rel = row[ROW_RHS];
            if (row[ROW_HASROB] != 0.f) {
#pragma unroll
                for (int d = 0; d < NJ; d++) { denom += row[d] * row[LY::R_OFF + d]; rel += row[d] * L.qd[d]; }
            }
            float dinv = denom > SIMD_EPS ? 1.f / denom : 0.f;
            row[ROW_DINV] = dinv;
            row[ROW_APP] = 0.f;
            if (t == 0) {
                float pos_err = 0.f, vel_err = -rel;
                if (dist > 0.f) vel_err -= dist / DT;
                else pos_err = -dist * CONTACT_ERP / DT;
                row[ROW_RHS] = (pos_err + vel_err) * dinv;
                row[ROW_MU] = 0.f;
            } else {
                row[ROW_RHS] = -rel * dinv;
                row[ROW_MU] = L.con_mu[c];
            }
        }
    }
    wv::lds_sync();
    PMG_PH(5);
}

/* ---------------------------------------------------------------- */
/* Gauss-Seidel visit of one LDS row (kernels with free bodies).  All row scalars are one 16-byte
 * uniform LDS read, J and the response are per-lane LDS reads, the accumulated impulse goes back
 * with one store: no v_readlane / SGPR round trips on the critical path of the solve, which is
 * VALU-issue bound when every env of the batch is in contact.                                  */
template <int NB>
struct LaneDof { /* which DoF of the constraint space this lane holds during the solve */
    int blk, comp;   /* block index / component (0..5), or -1 */
    float scale;     /* response scale of a block DoF: 1/m (linear) or 1/I (angular) */
};
template <int NB>
__device__ __forceinline__ void lane_dof(LaneDof<NB>& d)
{
    int l = wv::lane();
    d.blk = -1; d.comp = -1; d.scale = 0.f;
    if (l >= NJ && l < NJ + 6 * NB) {
        d.blk = (l - NJ) / 6;
        d.comp = (l - NJ) % 6;
        d.scale = d.comp < 3 ? 1.f / BLOCK_MASS : 1.f / BLOCK_INERTIA;
    }
}

/* what one visit needs from LDS; fetched one visit AHEAD (software pipelining): none of it depends on
 * the running delta-velocity, so the ~100-cycle LDS latency hides behind the previous row's reduction */
struct RowData {
    float J, resp, rhs, dinv, app;
};
template <int NB>
__device__ __forceinline__ void row_fetch(const float* row, const LaneDof<NB>& ld, RowData& d)
{
    using LY = RowLayout<NB>;
    int l = wv::lane();
    if (LY::direct) {
        d.J = l < 16 ? row[l] : 0.f;
        d.resp = l < 16 ? row[16 + l] : 0.f;
    } else {
        int ida = (int)row[SLOT_IDA], idb = (int)row[SLOT_IDB];
        int slot = l < NJ ? l : (ld.blk >= 0 ? (ld.blk == ida ? 9 + ld.comp : (ld.blk == idb ? 15 + ld.comp : -1)) : -1);
        d.J = slot >= 0 ? row[slot] : 0.f;
        d.resp = l < NJ ? row[LY::R_OFF + l] : d.J * ld.scale;
    }
    d.rhs = row[ROW_RHS];
    d.dinv = row[ROW_DINV];
    d.app = row[ROW_APP];
}

/* K = 0 normal row (bounds [0, 1e10]), K = 1 friction row (bounds +-lim, lim wave-uniform).
 * dv: this lane's delta-velocity DoF; returns the squared velocity change (valid in DoF lanes). */
template <int NB, int K>
__device__ __forceinline__ float lds_row_solve(float* row, const RowData& rd, float lim, float& dv)
{
    using LY = RowLayout<NB>;
    int l = wv::lane();
    float x = rd.J * dv;
    float jd;
    if (LY::direct) jd = wv::row_sum(x);                 /* every DoF lives in lanes 0..14: row-local, no broadcast */
    else jd = wv::sum_rows<(NJ + 6 * NB + 15) / 16>(x);
    float sum = rd.app + (rd.rhs - jd * rd.dinv);
    float napp = K == 0 ? fminf(fmaxf(sum, 0.f), 1e10f) : __builtin_amdgcn_fmed3f(sum, -lim, lim);
    float delta = napp - rd.app;
    if (l == 0) row[ROW_APP] = napp;
    dv += rd.resp * delta;
    float d = rd.dinv != 0.f ? delta / rd.dinv : 0.f;
    if (LY::direct && l >= 16) d = 0.f; /* only row 0 computed the real impulse */
    return d * d;
}

/* one PGS iteration over the LDS contact rows: normals in list order, then the friction pairs of
 * the contacts that carry a positive normal impulse ([BULLET-PRIOR] solveSingleIteration).
 * `loaded` (wave-uniform bit mask) records which normals ended the pass with a positive impulse,
 * so the friction pass branches on scalar bits instead of LDS round trips. */
template <int NB, int MAXC>
__device__ __forceinline__ float lds_rows_iteration(ContactLds<NB, MAXC>& L, int nc, const LaneDof<NB>& ld, float& dv)
{
    float resid = 0.f;
    unsigned long long loaded = 0ull;
    RowData cur, nxt;
    row_fetch<NB>(L.rows[0], ld, cur);
    for (int cc = 0; cc < nc; cc++) {
        row_fetch<NB>(L.rows[cc + 1 < nc ? cc + 1 : cc], ld, nxt);
        resid = fmaxf(resid, lds_row_solve<NB, 0>(L.rows[cc], cur, 0.f, dv));
        cur = nxt;
    }
    /* which normals carry an impulse now: one uniform LDS read each, folded into a scalar mask */
    for (int cc = 0; cc < nc; cc++)
        if (wv::uniform_positive(L.rows[cc][ROW_APP])) loaded |= 1ull << cc;
    while (loaded) {
        int cc = __builtin_ctzll(loaded);
        loaded &= loaded - 1ull;
        float* r1 = L.rows[MAXC + 2 * cc];
        float lim = r1[ROW_MU] * L.rows[cc][ROW_APP];
        RowData a, b;
        row_fetch<NB>(r1, ld, a);
        row_fetch<NB>(r1 + ROW_STRIDE, ld, b);
        resid = fmaxf(resid, lds_row_solve<NB, 1>(r1, a, lim, dv));
        resid = fmaxf(resid, lds_row_solve<NB, 1>(r1 + ROW_STRIDE, b, lim, dv));
    }
    return resid;
}

/* ---------------------------------------------------------------- */
/* Reach kernel (no free bodies): every contact row is finger x table and involves only the
 * 9 robot DoFs in lanes 0..8.  With at most 8 contacts (2 fingers x 4 points) the whole row set
 * fits in registers: lane k keeps J[r][k] and (M^-1 J^T)[r][k] of all 24 rows, the row scalars are
 * replicated in every lane, and a visit is one multiply, one DPP butterfly and one fused update
 * -- no LDS traffic and no scalar broadcasts inside the five solver iterations.             */
template <int MAXC>
struct RobotRows {
    float J[3 * MAXC], R[3 * MAXC];          /* per lane (DoF) */
    float rhs[3 * MAXC], dinv[3 * MAXC], app[3 * MAXC], mu[MAXC]; /* replicated */
};

template <int NB, int MAXC>
__device__ __forceinline__ void load_robot_rows(ContactLds<NB, MAXC>& L, int nc, RobotRows<MAXC>& rr)
{
    int l = wv::lane();
    int k = l < NJ ? l : 0;
#pragma unroll
    for (int c = 0; c < MAXC; c++) {
#pragma unroll
        for (int t = 0; t < 3; t++) {
            const float* row = t == 0 ? L.rows[c] : L.rows[MAXC + 2 * c + (t - 1)];
            bool ok = c < nc;
            rr.J[3 * c + t] = (ok && l < NJ) ? row[k] : 0.f;
            rr.R[3 * c + t] = (ok && l < NJ) ? row[RowLayout<NB>::R_OFF + k] : 0.f;
            rr.rhs[3 * c + t] = ok ? row[ROW_RHS] : 0.f;
            rr.dinv[3 * c + t] = ok ? row[ROW_DINV] : 0.f;
            rr.app[3 * c + t] = 0.f;
        }
        rr.mu[c] = c < nc ? L.rows[MAXC + 2 * c][ROW_MU] : 0.f;
    }
}

/* one PGS iteration over the contact rows (normals, then frictions of loaded normals) */
template <int MAXC>
__device__ __forceinline__ float robot_rows_iteration(RobotRows<MAXC>& rr, int nc, float& dv)
{
    float resid = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; c++) {
        if (c < nc) {
            const int r = 3 * c;
            float jd = wv::sum_row0(rr.J[r] * dv);
            float napp = fmaxf(rr.app[r] + (rr.rhs[r] - jd * rr.dinv[r]), 0.f);
            napp = fminf(napp, 1e10f);
            float delta = napp - rr.app[r];
            rr.app[r] = napp;
            dv += rr.R[r] * delta;
            float d = rr.dinv[r] != 0.f ? delta / rr.dinv[r] : 0.f;
            resid = fmaxf(resid, d * d);
        }
    }
#pragma unroll
    for (int c = 0; c < MAXC; c++) {
        if (c < nc && rr.app[3 * c] > 0.f) {
            float lim = rr.mu[c] * rr.app[3 * c];
#pragma unroll
            for (int t = 1; t < 3; t++) {
                const int r = 3 * c + t;
                float jd = wv::sum_row0(rr.J[r] * dv);
                float napp = __builtin_amdgcn_fmed3f(rr.app[r] + (rr.rhs[r] - jd * rr.dinv[r]), -lim, lim);
                float delta = napp - rr.app[r];
                rr.app[r] = napp;
                dv += rr.R[r] * delta;
                float d = rr.dinv[r] != 0.f ? delta / rr.dinv[r] : 0.f;
                resid = fmaxf(resid, d * d);
            }
        }
    }
    return resid;
}

}  // namespace pmg
#endif
