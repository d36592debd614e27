// hoic_collide.h — narrow phase, one collision pair per lane.
//
// Replaces MuJoCo's collision stage for the static pair list of the HOIC models (the reference reads the
// result as data.contact[], uhc/envs/ho_im4.py:884-889).  Contacts follow MuJoCo's convention: dist < 0 on
// penetration, pos midway between the surfaces, normal from geom1 to geom2.  Same geometry as the CPU oracle
// (oracle/ho_collide.c) in float32; MuJoCo's exact contact multiset is not reproduced.
#pragma once
#include "hoic_types.h"
#include "hoic_math.h"

// Contacts found by one lane (= one pair): staged in LDS (Work::col_lc, one column per lane, COLSLOT contacts; a pair that
// can produce more -- plane-box, box-box, box-mesh, plane-mesh: DevModel::pair_pool -- owns a pool entry for contacts 2 and 3),
// so that the narrow phase holds neither 28 more registers nor a scratch-memory copy (dynamic indexing of a register array
// would put it there; scratch round trips go to L2/HBM and used to dominate this stage).
struct LaneContacts {
  int n;
  float* b;   // &col_lc[lane]
  float* b2;  // the pair's pool entry (any entry when cap == COLSLOT: never written)
  int cap;    // contacts this lane can stage: COLSLOT, or COLSLOT + 2 with a pool entry
};
HD float& lc_at(const LaneContacts& o, int q, int k) { return q < COLSLOT ? o.b[(q * 7 + k) * NT] : o.b2[(q - COLSLOT) * 7 + k]; }
HD void lc_put(LaneContacts& o, int slot, float dist, const float* pos, const float* n) {
  if (slot < 0 || slot >= o.cap) return;      // (build_model gives every pair type that can produce more than COLSLOT contacts a pool entry)
  lc_at(o, slot, 0) = dist;
  for (int i = 0; i < 3; i++) { lc_at(o, slot, 1 + i) = pos[i]; lc_at(o, slot, 4 + i) = n[i]; }
}
HD void lc_push(LaneContacts& o, float dist, const float* pos, const float* n) { if (o.n < o.cap) { lc_put(o, o.n, dist, pos, n); o.n++; } }

HD void col_plane_sphere(const float* pp, const float* pn, const float* c, float r, LaneContacts& o) {
  float d[3] = {c[0] - pp[0], c[1] - pp[1], c[2] - pp[2]};
  float dist = dot3(d, pn) - r;
  if (dist >= 0.f) return;
  float pos[3];
  for (int i = 0; i < 3; i++) pos[i] = c[i] - pn[i] * (r + 0.5f * dist);
  lc_push(o, dist, pos, pn);
}
HD void col_plane_capsule(const float* pp, const float* pR, const float* cp, const float* cR, const float* size, LaneContacts& o) {
  float n[3], ax[3], e[3];
  matcol(pR, 2, n); matcol(cR, 2, ax);
  for (int s = -1; s <= 1; s += 2) {
    for (int i = 0; i < 3; i++) e[i] = cp[i] + s * size[1] * ax[i];
    col_plane_sphere(pp, n, e, size[0], o);
  }
}
HD void col_plane_box(const float* pp, const float* pR, const float* bp, const float* bR, const float* h, LaneContacts& o) {
  float n[3];
  matcol(pR, 2, n);
  for (int k = 0; k < 8 && o.n < 4; k++) {
    float loc[3] = {(k & 1 ? h[0] : -h[0]), (k & 2 ? h[1] : -h[1]), (k & 4 ? h[2] : -h[2])}, wv[3], d[3];
    matvec(bR, loc, wv);
    for (int i = 0; i < 3; i++) { wv[i] += bp[i]; d[i] = wv[i] - pp[i]; }
    float dist = dot3(d, n);
    if (dist >= 0.f) continue;
    float pos[3];
    for (int i = 0; i < 3; i++) pos[i] = wv[i] - 0.5f * dist * n[i];
    lc_push(o, dist, pos, n);
  }
}

HD void seg_seg_closest(const float* p1, const float* d1, const float* p2, const float* d2, float& s_out, float& t_out) {
  float r[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]};
  float a = dot3(d1, d1), e = dot3(d2, d2), f = dot3(d2, r), s, t;
  if (a <= 1e-12f && e <= 1e-12f) { s_out = t_out = 0.f; return; }
  if (a <= 1e-12f) { s = 0.f; t = fminf(fmaxf(fdiv(f, e), 0.f), 1.f); }
  else {
    float c = dot3(d1, r);
    if (e <= 1e-12f) { t = 0.f; s = fminf(fmaxf(fdiv(-c, a), 0.f), 1.f); }
    else {
      float b = dot3(d1, d2), den = a * e - b * b;
      s = den > 1e-6f * a * e ? fminf(fmaxf(fdiv(b * f - c * e, den), 0.f), 1.f) : 0.5f;
      t = fdiv(b * s + f, e);
      if (t < 0.f) { t = 0.f; s = fminf(fmaxf(fdiv(-c, a), 0.f), 1.f); }
      else if (t > 1.f) { t = 1.f; s = fminf(fmaxf(fdiv(b - c, a), 0.f), 1.f); }
    }
  }
  s_out = s; t_out = t;
}
HD void col_capsule_capsule(const float* p1, const float* R1, const float* s1, const float* p2, const float* R2,
                            const float* s2, LaneContacts& o) {
  float a1[3], a2[3], q1[3], q2[3], d1[3], d2[3], s, t, c1[3], c2[3];
  matcol(R1, 2, a1); matcol(R2, 2, a2);
  for (int i = 0; i < 3; i++) {
    q1[i] = p1[i] - s1[1] * a1[i]; d1[i] = 2.f * s1[1] * a1[i];
    q2[i] = p2[i] - s2[1] * a2[i]; d2[i] = 2.f * s2[1] * a2[i];
  }
  seg_seg_closest(q1, d1, q2, d2, s, t);
  for (int i = 0; i < 3; i++) { c1[i] = q1[i] + s * d1[i]; c2[i] = q2[i] + t * d2[i]; }
  float d[3] = {c2[0] - c1[0], c2[1] - c1[1], c2[2] - c1[2]};
  float len = fsqrt(dot3(d, d)), dist = len - s1[0] - s2[0];
  if (dist >= 0.f) return;
  if (len < 1e-12f) { d[0] = 1.f; d[1] = d[2] = 0.f; } else { float inv = frcp(len); d[0] *= inv; d[1] *= inv; d[2] *= inv; }
  float pos[3];
  for (int i = 0; i < 3; i++) pos[i] = c1[i] + d[i] * (s1[0] + 0.5f * dist);
  lc_push(o, dist, pos, d);
}

// sphere (centre c in the box frame) vs box; outputs in the box frame
HD bool sphere_box_local(const float* c, float r, const float* h, float& dist, float* pos, float* n) {
  float q[3], d[3];
  bool inside = true;
  for (int i = 0; i < 3; i++) {
    q[i] = fminf(fmaxf(c[i], -h[i]), h[i]);
    d[i] = c[i] - q[i];
    if (d[i] != 0.f) inside = false;
  }
  if (!inside) {
    float len = fsqrt(dot3(d, d));
    dist = len - r;
    if (dist >= 0.f) return false;
    float inv = frcp(len);
    for (int i = 0; i < 3; i++) { n[i] = -d[i] * inv; pos[i] = q[i] + d[i] * inv * 0.5f * dist; }
    return true;
  }
  int k = 0; float best = 1e30f;
  for (int i = 0; i < 3; i++) { float dep = h[i] - fabsf(c[i]); if (dep < best) { best = dep; k = i; } }
  float sg = c[k] >= 0.f ? 1.f : -1.f;
  dist = -(best + r);
  for (int i = 0; i < 3; i++) { n[i] = 0.f; pos[i] = c[i]; }
  n[k] = -sg;
  pos[k] = c[k] + sg * 0.5f * (best - r);
  return true;
}
HD float seg_box_t(const float* a, const float* b, const float* h) {
  float t = 0.5f, lo = 0.f, hi = 1.f;
  for (int it = 0; it < 24; it++) {
    float g = 0.f, hh = 0.f;
    for (int i = 0; i < 3; i++) {
      float v = b[i] - a[i], p = a[i] + t * v;
      float ex = p > h[i] ? p - h[i] : (p < -h[i] ? p + h[i] : 0.f);
      if (ex != 0.f) { g += 2.f * ex * v; hh += 2.f * v * v; }
    }
    if (g > 0.f) hi = t; else if (g < 0.f) lo = t; else break;
    float tn = hh > 0.f ? t - fdiv(g, hh) : 0.5f * (lo + hi);
    if (tn <= lo || tn >= hi) tn = 0.5f * (lo + hi);
    if (fabsf(tn - t) < 1e-7f) { t = tn; break; }
    t = tn;
    if (hi - lo < 1e-7f) break;
  }
  return t;
}
HD void col_capsule_box(const float* cp, const float* cR, const float* cs, const float* bp, const float* bR,
                        const float* h, LaneContacts& o) {
  float ax[3], rel[3], pc[3], al[3], a[3], b[3];
  matcol(cR, 2, ax);
  for (int i = 0; i < 3; i++) rel[i] = cp[i] - bp[i];
  mattvec(bR, rel, pc); mattvec(bR, ax, al);
  for (int i = 0; i < 3; i++) { a[i] = pc[i] - cs[1] * al[i]; b[i] = pc[i] + cs[1] * al[i]; }
  const float r = cs[0];
  float dist, pos[3], n[3];
  // candidate axis parameters in the order of the sequential list: the two ends if they touch, else the closest point
  const bool h0 = sphere_box_local(a, r, h, dist, pos, n), h1 = sphere_box_local(b, r, h, dist, pos, n);
  float tc[3] = {0.f, 1.f, 0.f};
  bool tv[3] = {h0, h1, false};
  if (!(h0 && h1)) {
    const float ts = seg_box_t(a, b, h);
    tc[2] = ts;
    tv[2] = !((h0 && fabsf(ts) < 1e-4f) || (h1 && fabsf(ts - 1.f) < 1e-4f));
  }
  int cnt = 0;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    if (!tv[k] || cnt >= 2) continue;
    float c[3];
    for (int i = 0; i < 3; i++) c[i] = a[i] + tc[k] * (b[i] - a[i]);
    if (!sphere_box_local(c, r, h, dist, pos, n)) continue;
    float pw[3], nw[3];
    matvec(bR, pos, pw); matvec(bR, n, nw);
    for (int i = 0; i < 3; i++) pw[i] += bp[i];
    lc_push(o, dist, pw, nw);
    cnt++;
  }
}

// ---- box-box, wave-cooperative: the same separating-axis test / reference-face clipping as col_box_box (and as
// oracle/ho_collide.c box_box), with the fifteen axes on fifteen lanes and the clipped polygon one vertex per lane.
// The per-lane version walks ~3000 dependent instructions over scratch-resident polygons on ONE lane while 63
// idle; this one is a few hundred wave-wide steps.  All inputs are wave-uniform (every lane passes the same pair);
// decisions keep the sequential tie-breaks (first axis with the smallest penetration, polygon order).
// Returns the number of contacts n <= 4 (wave-uniform) and writes them into the owner lane's staging column `o`.
// polybuf (LDS): 32 floats of staging for the clipping.
__device__ __forceinline__ int col_box_box_wave(const float* pa, const float* Ra, const float* ha, const float* pb, const float* Rb,
                                                const float* hb, const LaneContacts& o, float* polybuf) {
  const int lane = opaque(threadIdx.x);
  float A[3][3], B[3][3], R[3][3], Q[3][3], t[3], tw[3];
  for (int i = 0; i < 3; i++) { matcol(Ra, i, A[i]); matcol(Rb, i, B[i]); tw[i] = pb[i] - pa[i]; }
  for (int i = 0; i < 3; i++) {
    t[i] = dot3(tw, A[i]);
    for (int j = 0; j < 3; j++) { R[i][j] = dot3(A[i], B[j]); Q[i][j] = fabsf(R[i][j]) + 1e-6f; }
  }
  // ---- separating axes, lane = axis: 0-2 faces of A, 3-5 faces of B, 6-14 edge pairs (i, j) = ((lane-6)/3, (lane-6)%3)
  float pen = 1e30f, nrm[3] = {0.f, 0.f, 0.f};
  bool valid = false;
  if (lane < 3) {
    const int i = lane;
    const float ti = sel3(t[0], t[1], t[2], i);
    const float rb = hb[0] * sel3(Q[0][0], Q[1][0], Q[2][0], i) + hb[1] * sel3(Q[0][1], Q[1][1], Q[2][1], i) + hb[2] * sel3(Q[0][2], Q[1][2], Q[2][2], i);
    pen = sel3(ha[0], ha[1], ha[2], i) + rb - fabsf(ti);
    const float sg = ti < 0.f ? -1.f : 1.f;
    float Ai[3]; sel3v(A, i, Ai);
    for (int k = 0; k < 3; k++) nrm[k] = sg * Ai[k];
    valid = true;
  } else if (lane < 6) {
    const int j = lane - 3;
    const float tb = t[0] * sel3(R[0][0], R[0][1], R[0][2], j) + t[1] * sel3(R[1][0], R[1][1], R[1][2], j) + t[2] * sel3(R[2][0], R[2][1], R[2][2], j);
    const float ra = ha[0] * sel3(Q[0][0], Q[0][1], Q[0][2], j) + ha[1] * sel3(Q[1][0], Q[1][1], Q[1][2], j) + ha[2] * sel3(Q[2][0], Q[2][1], Q[2][2], j);
    pen = ra + sel3(hb[0], hb[1], hb[2], j) - fabsf(tb);
    const float sg = tb < 0.f ? -1.f : 1.f;
    float Bj[3]; sel3v(B, j, Bj);
    for (int k = 0; k < 3; k++) nrm[k] = sg * Bj[k];
    valid = true;
  } else if (lane < 15) {
    const int i = (lane - 6) / 3, j = (lane - 6) % 3;
    float Ai[3], Bj[3], L[3];
    sel3v(A, i, Ai); sel3v(B, j, Bj);
    cross3(Ai, Bj, L);
    const float len = fsqrt(dot3(L, L));
    if (len >= 1e-4f) {
      const float inv = frcp(len);
      for (int k = 0; k < 3; k++) L[k] *= inv;
      float ra = 0.f, rb = 0.f;
      for (int k = 0; k < 3; k++) { ra += ha[k] * fabsf(dot3(A[k], L)); rb += hb[k] * fabsf(dot3(B[k], L)); }
      const float tl = dot3(tw, L);
      pen = ra + rb - fabsf(tl);
      const float sg = tl < 0.f ? -1.f : 1.f;
      for (int k = 0; k < 3; k++) nrm[k] = sg * L[k];
      valid = true;
    }
  }
  if (__ballot(valid && pen < 0.f) != 0ull) return 0;
  const float best = wave_min(lane < 6 ? pen : 1e30f);
  const int code = __ffsll((long long)__ballot(lane < 6 && pen == best)) - 1;
  const bool isedge = lane >= 6 && valid;
  const float beste = wave_min(isedge ? pen : 1e30f);
  const unsigned long long emask = __ballot(isedge && pen == beste);
  const int ecode = emask ? (__ffsll((long long)emask) - 1 - 6) : -1;
  if (ecode >= 0 && beste * 1.05f + 1e-6f < best) {      // ---- edge-edge: one contact (uniform code on every lane)
    const int i = ecode / 3, j = ecode % 3;
    float en[3];
    for (int k = 0; k < 3; k++) en[k] = rl(nrm[k], 6 + ecode);
    float ea[3], eb[3];
    for (int k = 0; k < 3; k++) { ea[k] = pa[k]; eb[k] = pb[k]; }
    for (int k = 0; k < 3; k++) {
      if (k != i) { const float s = dot3(en, A[k]) > 0.f ? 1.f : -1.f; for (int c = 0; c < 3; c++) ea[c] += s * ha[k] * A[k][c]; }
      if (k != j) { const float s = dot3(en, B[k]) > 0.f ? -1.f : 1.f; for (int c = 0; c < 3; c++) eb[c] += s * hb[k] * B[k][c]; }
    }
    float Ai[3], Bj[3];
    sel3v(A, i, Ai); sel3v(B, j, Bj);
    const float r[3] = {ea[0] - eb[0], ea[1] - eb[1], ea[2] - eb[2]};
    const float bdot = dot3(Ai, Bj), c1 = dot3(Ai, r), f1 = dot3(Bj, r), den = 1.f - bdot * bdot;
    float u = den > 1e-6f ? fdiv(bdot * f1 - c1, den) : 0.f, v = f1 + bdot * u;
    const float hai = sel3(ha[0], ha[1], ha[2], i), hbj = sel3(hb[0], hb[1], hb[2], j);
    u = fminf(fmaxf(u, -hai), hai); v = fminf(fmaxf(v, -hbj), hbj);
    if (lane == 0) {
      lc_at(o, 0, 0) = -beste;
      for (int k = 0; k < 3; k++) { lc_at(o, 0, 1 + k) = 0.5f * (ea[k] + u * Ai[k] + eb[k] + v * Bj[k]); lc_at(o, 0, 4 + k) = en[k]; }
    }
    return 1;
  }
  // ---- face contact: clip the incident face against the side planes of the reference face
  float bestn[3];
  for (int k = 0; k < 3; k++) bestn[k] = rl(nrm[k], code);
  const bool refA = code < 3; const int ax = refA ? code : code - 3;
  float pr[3], pi_[3], hr[3], hi[3], Rr[3][3], Ri[3][3], nref[3];
  for (int k = 0; k < 3; k++) {
    pr[k] = refA ? pa[k] : pb[k]; pi_[k] = refA ? pb[k] : pa[k]; hr[k] = refA ? ha[k] : hb[k]; hi[k] = refA ? hb[k] : ha[k];
    nref[k] = refA ? bestn[k] : -bestn[k];
    for (int c = 0; c < 3; c++) { Rr[k][c] = refA ? A[k][c] : B[k][c]; Ri[k][c] = refA ? B[k][c] : A[k][c]; }
  }
  int iax = 0; float mind = 1e30f, isg = 1.f;
  for (int k = 0; k < 3; k++) {
    const float dd = dot3(Ri[k], nref);
    if (-fabsf(dd) < mind) { mind = -fabsf(dd); iax = k; isg = dd > 0.f ? -1.f : 1.f; }
  }
  const int i1 = (iax + 1) % 3, i2 = (iax + 2) % 3, r1 = (ax + 1) % 3, r2 = (ax + 2) % 3;
  float Riax[3], Ri1[3], Ri2[3], Rr1[3], Rr2[3];
  sel3v(Ri, iax, Riax); sel3v(Ri, i1, Ri1); sel3v(Ri, i2, Ri2); sel3v(Rr, r1, Rr1); sel3v(Rr, r2, Rr2);
  const float hiax = sel3(hi[0], hi[1], hi[2], iax), hi1 = sel3(hi[0], hi[1], hi[2], i1), hi2 = sel3(hi[0], hi[1], hi[2], i2);
  const float hr1 = sel3(hr[0], hr[1], hr[2], r1), hr2 = sel3(hr[0], hr[1], hr[2], r2), hrax = sel3(hr[0], hr[1], hr[2], ax);
  float fc[3];
  for (int k = 0; k < 3; k++) fc[k] = pi_[k] + isg * hiax * Riax[k] - pr[k];
  // incident-face corners: lane v in 0..3 (every lane computes corner lane & 3; lanes >= 4 are ignored)
  float px, py, vz;
  {
    const int v = lane & 3;
    const float sx = (v == 0 || v == 3) ? 1.f : -1.f, sy = (v < 2) ? 1.f : -1.f;   // (1,1) (-1,1) (-1,-1) (1,-1)
    float wv[3];
    for (int k = 0; k < 3; k++) wv[k] = fc[k] + sx * hi1 * Ri1[k] + sy * hi2 * Ri2[k];
    px = dot3(wv, Rr1); py = dot3(wv, Rr2); vz = dot3(wv, nref);
  }
  const float p0x = rl(px, 0), p0y = rl(py, 0), z_0 = rl(vz, 0);
  const float m00 = rl(px, 1) - p0x, m01 = rl(py, 1) - p0y, m10 = rl(px, 3) - p0x, m11 = rl(py, 3) - p0y;
  const float det = m00 * m11 - m01 * m10;
  float gx = 0.f, gy = 0.f;
  if (fabsf(det) > 1e-12f) {
    const float dz1 = rl(vz, 1) - z_0, dz3 = rl(vz, 3) - z_0;
    const float idet = frcp(det); gx = (dz1 * m11 - dz3 * m01) * idet; gy = (dz3 * m00 - dz1 * m10) * idet;
  }
  const float z0 = z_0 - gx * p0x - gy * p0y;
  int n = 4;
#pragma unroll
  for (int stage = 0; stage < 4; stage++) {
    const int axis = stage >> 1; const float lim = axis ? hr2 : hr1, sgn = (stage & 1) ? -1.f : 1.f;
    const int nxt = (lane + 1 < n) ? lane + 1 : 0;
    const float bx = __shfl(px, nxt), by = __shfl(py, nxt);
    const bool act = lane < n;
    const float da = sgn * (axis ? py : px) - lim, db = sgn * (axis ? by : bx) - lim;
    const bool e1 = act && da <= 0.f, e2 = act && ((da < 0.f && db > 0.f) || (da > 0.f && db < 0.f));
    const int cnt = (e1 ? 1 : 0) + (e2 ? 1 : 0);
    const int incl = wave_incl_scan(cnt);
    const int pos = incl - cnt;
    n = __builtin_amdgcn_readlane(incl, NT - 1);
    if (n == 0) return 0;
    if (e1) { polybuf[2 * pos] = px; polybuf[2 * pos + 1] = py; }
    if (e2) {
      const float tt = fdiv(da, da - db);
      polybuf[2 * (pos + (e1 ? 1 : 0))] = px + tt * (bx - px); polybuf[2 * (pos + (e1 ? 1 : 0)) + 1] = py + tt * (by - py);
    }
    wsync();
    if (lane < n) { px = polybuf[2 * lane]; py = polybuf[2 * lane + 1]; }
    wsync();
  }
  // depths, keep the penetrating vertices (polygon order), at most four of them
  const float depth = hrax - (z0 + gx * px + gy * py);
  const unsigned long long kmask = __ballot(lane < n && depth > 0.f);
  const int nk = __popcll(kmask);
  if (nk == 0) return 0;
  const int krank = __popcll(kmask & ((1ull << lane) - 1ull));    // position of this lane's vertex in keep[]
  const bool kept = (kmask >> lane) & 1ull;
  int srank = -1;                                                    // output slot of this lane's vertex (-1: dropped)
  int ns = nk;
  if (nk <= 4) { if (kept) srank = krank; }
  else {
    const float dmax = wave_max(kept ? depth : -1e30f);
    const int d0lane = __ffsll((long long)__ballot(kept && depth == dmax)) - 1;       // first deepest vertex
    const int d0 = __popcll(kmask & ((1ull << d0lane) - 1ull));
    ns = 4;
    for (int k = 0; k < 4; k++) if (kept && krank == (d0 + (k * nk) / 4) % nk) srank = k;
  }
  if (srank >= 0) {
    const float z = hrax - depth;
    lc_at(o, srank, 0) = -depth;
    for (int k = 0; k < 3; k++) { lc_at(o, srank, 1 + k) = pr[k] + px * Rr1[k] + py * Rr2[k] + (z + 0.5f * depth) * nref[k]; lc_at(o, srank, 4 + k) = bestn[k]; }
  }
  return ns;
}

// ---- convex mesh (hull vertices + face planes in the geom frame) vs plane / capsule / box, wave-cooperative: the
// whole wave works on one pair at a time (like box-box).  The hull tables stay in global memory (L1 / L2 resident, a few
// tens of KB per model) and are streamed 64 entries per pass, one per lane, so a hull may have any size up to the
// compiled capacities (the reference's full hulls: 130 / 258 vertices for the bottle, 231 / 707 / 939 for the banana).
// "The face of largest signed distance at a point" - the kernel of all three routines - is one coalesced float4 load
// and 4 FMAs per lane and pass plus one wave maximum, and the vertex loops run one vertex per lane.  The logic around it
// is the sequential formulation of oracle/ho_collide.c, evaluated uniformly by all lanes; results that depend on an
// order (first maximum, keep-the-deepest-four, three-lowest) are produced in that order.
typedef float f4v __attribute__((ext_vector_type(4)));
// Exact pruning.  The tables are ordered so that runs of HOIC_HULL_RUN_VERTS vertices are compact and runs of
// HOIC_HULL_RUN_FACES faces have similar normals (hoic_amd/mjcf.py coherent_order); build_model bounds every run:
//   vertex run: bounding sphere (centre, radius)  -> a run none of whose vertices can qualify is not loaded;
//   face run:   n.x - d <= sum_i max(nlo_i w_i, nhi_i w_i) + emax with w = x - c   -> a run whose bound is below a value
//               already found cannot hold the maximum (nor tie with it) and is not loaded.
// What is loaded is evaluated with the same arithmetic and in the same table order as the streaming form (prune == 0,
// HOIC_MESH_STREAM=1), so both forms produce bit-identical contacts (tests/test_gpu_parity.py
// test_mesh_pruning_changes_no_contact); the win is memory traffic: a query on the banana's largest hull reads ~4 KB
// instead of 29 KB of plane rows through the CU's vector L1, which is what bounded the mesh configurations.
#ifndef HULL_STREAM_BELOW
#define HULL_STREAM_BELOW 512
#endif
struct HullRef { GPTR(const f4v) pl; GPTR(const f4v) vv; GPTR(const f4v) vrun; GPTR(const f4v) frun; int np, nv, nvr, nfr, prune; float lo[3], hi[3]; };
HD HullRef hull_ref(const DevModel& m, int mesh) {
  HullRef h;
  h.np = m.mesh_planenum[mesh]; h.nv = m.mesh_vertnum[mesh]; h.nvr = m.mesh_vrunnum[mesh]; h.nfr = m.mesh_frunnum[mesh]; h.prune = m.mesh_prune;
  h.pl = (GPTR(const f4v))(GPTR(const void))&m.mesh_plane[m.mesh_planeadr[mesh]][0];
  h.vv = (GPTR(const f4v))(GPTR(const void))&m.mesh_vert[m.mesh_vertadr[mesh]][0];
  h.vrun = (GPTR(const f4v))(GPTR(const void))&m.mesh_vrun[m.mesh_vrunadr[mesh]][0];
  h.frun = (GPTR(const f4v))(GPTR(const void))&m.mesh_frun[m.mesh_frunadr[mesh]][0];
  for (int i = 0; i < 3; i++) { h.lo[i] = m.mesh_aabb[mesh][i]; h.hi[i] = m.mesh_aabb[mesh][4 + i]; }
  return h;
}
HD float hull_plane_val(const f4v p, float x, float y, float z) { return fmaf(p.x, x, fmaf(p.y, y, fmaf(p.z, z, -p.w))); }
// upper bound of n.x - d over face run r (lane-private r), with room for the rounding of this evaluation
HD float hull_run_bound(const HullRef& h, int r, float x, float y, float z) {
  const f4v c = h.frun[3 * r], lo = h.frun[3 * r + 1], hi = h.frun[3 * r + 2];
  const float wx = x - c.x, wy = y - c.y, wz = z - c.z;
  const float b = fmaxf(lo.x * wx, hi.x * wx) + fmaxf(lo.y * wy, hi.y * wy) + fmaxf(lo.z * wz, hi.z * wz) + c.w;
  return b + (2e-6f * (fabsf(wx) + fabsf(wy) + fabsf(wz) + fabsf(c.w)) + 1e-8f);
}
// pass 2 of a query: the runs whose bound (ub0: run = lane, ub1: run = lane + 64) can still reach `best` -- a value some face
// certainly attains at the point -- eight per trip (four per half-wave, all loads in flight) in ascending run order.  A lane's
// running best (bv, bi, bp) takes a face when it is larger, or equal with a smaller index: the selection "largest value, first
// index" does not depend on the order in which candidates are met, nor on rows fetched beyond need.
HD void hull_visit(const HullRef& h, float x, float y, float z, float ub0, float ub1, float best, float& bv, int& bi, f4v& bp) {
  const int lane = threadIdx.x, half = lane >> 5, sub = lane & 31;
#pragma unroll 1
  for (int part = 0; part < 2; part++) {
    if (part == 1 && h.nfr <= NT) break;
    unsigned long long mask = __ballot((part ? ub1 : ub0) >= best);
    while (mask) {
      int tq[4]; bool on[4]; f4v pq[4];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        int ra = -1, rb = -1;
        if (mask) { ra = __ffsll((long long)mask) - 1; mask &= mask - 1; }
        if (mask) { rb = __ffsll((long long)mask) - 1; mask &= mask - 1; }
        const int rr = half ? rb : ra;
        tq[q] = (rr + part * NT) * HOIC_HULL_RUN_FACES + sub;
        on[q] = rr >= 0 && tq[q] < h.np;
        pq[q] = h.pl[on[q] ? tq[q] : 0];
      }
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const float v = hull_plane_val(pq[q], x, y, z);
        if (on[q] && (v > bv || (v == bv && tq[q] < bi))) { bv = v; bi = tq[q]; bp = pq[q]; }
      }
      if (mask) {                                  // more candidates: raise the bar with what this trip found
        best = fmaxf(best, wave_max(bv));
        mask &= __ballot((part ? ub1 : ub0) >= best);
      }
    }
  }
}
// the wave's result from the lanes' running bests: the largest value, its first face; that face's plane comes out of the winning
// lane's registers (no table read)
HD float hull_select(const HullRef& h, float bv, int bi, const f4v bp, float* pl) {
  const float mx = wave_max(bv);
  const int idx = min((int)wave_min(bv == mx ? (float)bi : 1e9f), h.np - 1);     // indices < 2^24 are exact in float32 (the clamp: non-finite query points)
  const unsigned long long win = __ballot(bv == mx && bi == idx);
  if (win) {
    const int L = __ffsll((long long)win) - 1;
    pl[0] = rl(bp.x, L); pl[1] = rl(bp.y, L); pl[2] = rl(bp.z, L); pl[3] = rl(bp.w, L);
  } else {                                             // (non-finite query point: no lane compares equal)
    const f4v w = h.pl[idx];                           // wave-uniform address
    pl[0] = w.x; pl[1] = w.y; pl[2] = w.z; pl[3] = w.w;
  }
  return mx;
}
// max over the faces of n.x - d at the point (x, y, z) (hull frame); pl = that face (the first one in face order on ties)
// Round trips to the (L2-resident) tables are what a query costs -- a capsule pair makes two to five queries one after the other,
// each two to three dependent reads deep -- so: every lane keeps the plane of its best face in registers, the runs that pass 1
// leaves in play are fetched up to EIGHT per trip, the two ends of a capsule share their passes (hull_max_wave2) and the queries
// between them need no pass 1 at all (hull_max_wave_seg).
HD float hull_max_wave(const HullRef& h, float x, float y, float z, float* pl) {
  const int lane = threadIdx.x;
  if (h.np <= 0) { pl[0] = pl[1] = pl[2] = pl[3] = 0.f; return -1e30f; }      // a mesh without face planes (wave-uniform): no face, nothing to index
  float bv = -1e30f; int bi = 0x00ffffff;
  f4v bp = {0.f, 0.f, 0.f, 0.f};
  if (!h.prune || h.np <= HULL_STREAM_BELOW) {          // (small hulls: a handful of streaming passes cost less than bounds + visits)
    for (int t = lane; t < h.np; t += NT) {          // ascending index per lane: '>' keeps the lane's first maximum
      const f4v p = h.pl[t];
      const float v = hull_plane_val(p, x, y, z);
      if (v > bv) { bv = v; bi = t; bp = p; }
    }
  } else {
    // pass 1, lane = run (a second sub-pass for hulls of more than 64 runs): the run's upper bound, and its first face as a
    // value that is certainly attained
    float ub0 = -1e30f, ub1 = -1e30f;
    if (lane < h.nfr) {
      ub0 = hull_run_bound(h, lane, x, y, z);
      const f4v p = h.pl[lane * HOIC_HULL_RUN_FACES];
      bv = hull_plane_val(p, x, y, z); bi = lane * HOIC_HULL_RUN_FACES; bp = p;
    }
    if (h.nfr > NT && lane + NT < h.nfr) {
      ub1 = hull_run_bound(h, lane + NT, x, y, z);
      const f4v p = h.pl[(lane + NT) * HOIC_HULL_RUN_FACES];
      const float v = hull_plane_val(p, x, y, z);
      if (v > bv) { bv = v; bi = (lane + NT) * HOIC_HULL_RUN_FACES; bp = p; }
    }
    hull_visit(h, x, y, z, ub0, ub1, wave_max(bv), bv, bi, bp);
  }
  return hull_select(h, bv, bi, bp, pl);
}
// The same query at TWO points at once (the two ends of a capsule's axis: a few centimetres apart, so the runs in play are
// largely the same): one pass 1 (the run bounds' table reads are shared), and every face row fetched in pass 2 is evaluated at
// both points -- a row that only one point needed is harmless for the other (see above).  Results per point are those of
// hull_max_wave, bit for bit: the same faces' values, largest value, first index.  `sg` keeps the lanes' run bounds at the two
// points for the queries between them.
struct HullSeg { float ua0, ua1, ub0, ub1; };
HD void hull_max_wave2(const HullRef& h, const float* a, const float* b, float* pla, float* plb, float& va, float& vb, HullSeg& sg) {
  const int lane = threadIdx.x;
  sg.ua0 = sg.ua1 = sg.ub0 = sg.ub1 = -1e30f;
  if (h.np <= 0) { va = hull_max_wave(h, a[0], a[1], a[2], pla); vb = va; for (int i = 0; i < 4; i++) plb[i] = pla[i]; return; }
  float bva = -1e30f, bvb = -1e30f; int bia = 0x00ffffff, bib = 0x00ffffff;
  f4v bpa = {0.f, 0.f, 0.f, 0.f}, bpb = {0.f, 0.f, 0.f, 0.f};
  if (!h.prune || h.np <= HULL_STREAM_BELOW) {     // (wave-uniform: small hulls stream, every row evaluated at both points)
    for (int t = lane; t < h.np; t += NT) {
      const f4v p = h.pl[t];
      const float wa = hull_plane_val(p, a[0], a[1], a[2]), wb = hull_plane_val(p, b[0], b[1], b[2]);
      if (wa > bva) { bva = wa; bia = t; bpa = p; }
      if (wb > bvb) { bvb = wb; bib = t; bpb = p; }
    }
    va = hull_select(h, bva, bia, bpa, pla); vb = hull_select(h, bvb, bib, bpb, plb);
    return;
  }
#pragma unroll
  for (int part = 0; part < 2; part++) {
    const int r = lane + part * NT;
    if (part == 1 && h.nfr <= NT) break;
    if (r < h.nfr) {
      const float ua = hull_run_bound(h, r, a[0], a[1], a[2]), ub = hull_run_bound(h, r, b[0], b[1], b[2]);
      if (part) { sg.ua1 = ua; sg.ub1 = ub; } else { sg.ua0 = ua; sg.ub0 = ub; }
      const int t = r * HOIC_HULL_RUN_FACES;
      const f4v p = h.pl[t];
      const float wa = hull_plane_val(p, a[0], a[1], a[2]), wb = hull_plane_val(p, b[0], b[1], b[2]);
      if (wa > bva) { bva = wa; bia = t; bpa = p; }
      if (wb > bvb) { bvb = wb; bib = t; bpb = p; }
    }
  }
  float besta = wave_max(bva), bestb = wave_max(bvb);
  const int half = lane >> 5, sub = lane & 31;
#pragma unroll 1
  for (int part = 0; part < 2; part++) {
    if (part == 1 && h.nfr <= NT) break;
    unsigned long long mask = __ballot((part ? sg.ua1 : sg.ua0) >= besta || (part ? sg.ub1 : sg.ub0) >= bestb);
    while (mask) {
      int tq[4]; bool on[4]; f4v pq[4];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        int ra = -1, rb = -1;
        if (mask) { ra = __ffsll((long long)mask) - 1; mask &= mask - 1; }
        if (mask) { rb = __ffsll((long long)mask) - 1; mask &= mask - 1; }
        const int rr = half ? rb : ra;
        tq[q] = (rr + part * NT) * HOIC_HULL_RUN_FACES + sub;
        on[q] = rr >= 0 && tq[q] < h.np;
        pq[q] = h.pl[on[q] ? tq[q] : 0];
      }
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const float wa = hull_plane_val(pq[q], a[0], a[1], a[2]), wb = hull_plane_val(pq[q], b[0], b[1], b[2]);
        if (on[q] && (wa > bva || (wa == bva && tq[q] < bia))) { bva = wa; bia = tq[q]; bpa = pq[q]; }
        if (on[q] && (wb > bvb || (wb == bvb && tq[q] < bib))) { bvb = wb; bib = tq[q]; bpb = pq[q]; }
      }
      if (mask) {
        besta = fmaxf(besta, wave_max(bva)); bestb = fmaxf(bestb, wave_max(bvb));
        mask &= __ballot((part ? sg.ua1 : sg.ua0) >= besta || (part ? sg.ub1 : sg.ub0) >= bestb);
      }
    }
  }
  va = hull_select(h, bva, bia, bpa, pla); vb = hull_select(h, bvb, bib, bpb, plb);
}
// The query at a point BETWEEN the two of hull_max_wave2, x = a + t (b - a) with 0 <= t <= 1, without pass 1.  A run's bound --
// like the largest face value of the run that it bounds -- is a convex function of the point, so between the two points it lies
// below the chord of its values at them: (1 - t) ua + t ub, plus room for the rounding of the chord, of the point and of the
// face values themselves (1e-6: a micron, the bounds carry that much themselves), bounds the run at x with no table read.  The
// bar is the larger of the two faces `qa`, `qb` of this hull at x (the search's bracketing faces: their lines cross at x, so it is
// close to the answer) -- a value some face attains, which is all pass 2 needs: the maximum, and every face that ties with it,
// sits in a run whose bound reaches the bar.  Same face values, same selection: the result is that of hull_max_wave, bit for bit
// (test_mesh_pruning_changes_no_contact compares with the streaming form).
HD float hull_max_wave_seg(const HullRef& h, float x, float y, float z, float t, const HullSeg& sg, const float* qa, const float* qb, float* pl) {
  if (h.np <= 0 || !h.prune || h.np <= HULL_STREAM_BELOW) return hull_max_wave(h, x, y, z, pl);
  const f4v fa = {qa[0], qa[1], qa[2], qa[3]}, fb = {qb[0], qb[1], qb[2], qb[3]};
  const float bar = fmaxf(hull_plane_val(fa, x, y, z), hull_plane_val(fb, x, y, z));
  const float u0 = fmaf(t, sg.ub0 - sg.ua0, sg.ua0) + (1e-6f + 1e-6f * (fabsf(sg.ua0) + fabsf(sg.ub0)));
  const float u1 = fmaf(t, sg.ub1 - sg.ua1, sg.ua1) + (1e-6f + 1e-6f * (fabsf(sg.ua1) + fabsf(sg.ub1)));
  float bv = -1e30f; int bi = 0x00ffffff;
  f4v bp = {0.f, 0.f, 0.f, 0.f};
  hull_visit(h, x, y, z, u0, u1, bar, bv, bi, bp);
  return hull_select(h, bv, bi, bp, pl);
}
// the sequential "keep the four deepest" of one pair's contact list, state held uniformly: n, the four distances
struct Deep4 { int n; float d0, d1, d2, d3; };
HD void deep4_add(Deep4& k, const LaneContacts& owner, float dist, const float* pos, const float* nrm) {
  int slot = -1;
  if (k.n < 4) slot = k.n++;
  else {
    int wst = 0; float wd = k.d0;
    if (k.d1 > wd) { wd = k.d1; wst = 1; }
    if (k.d2 > wd) { wd = k.d2; wst = 2; }
    if (k.d3 > wd) { wd = k.d3; wst = 3; }
    if (dist < wd) slot = wst;
  }
  if (slot < 0) return;
  k.d0 = slot == 0 ? dist : k.d0; k.d1 = slot == 1 ? dist : k.d1; k.d2 = slot == 2 ? dist : k.d2; k.d3 = slot == 3 ? dist : k.d3;
  if (threadIdx.x == 0) { LaneContacts o = owner; lc_put(o, slot, dist, pos, nrm); }
}
__device__ __forceinline__ int col_plane_mesh_wave(const HullRef& h, const float* pp, const float* pR, const float* mp,
                                                const float* mR, const LaneContacts& owner) {
  const int lane = threadIdx.x;
  float n[3];
  matcol(pR, 2, n);
  float best[3] = {0.f, 0.f, 0.f}, bpos[3][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
  int nb = 0;
  // runs of 64 vertices whose bounding sphere reaches below the plane (lane = run); the others hold no candidate
  unsigned long long runs = ~0ull;
  if (h.prune) {
    bool reach = false;
    if (lane < h.nvr) {
      const f4v sp = h.vrun[lane];
      const float lc[3] = {sp.x, sp.y, sp.z};
      float wc[3];
      matvec(mR, lc, wc);
      const float dc = (wc[0] + mp[0] - pp[0]) * n[0] + (wc[1] + mp[1] - pp[1]) * n[1] + (wc[2] + mp[2] - pp[2]) * n[2];
      reach = dc - sp.w < 1e-6f * (1.f + fabsf(dc));
    }
    runs = __ballot(reach);
  }
  for (int base = 0; base < h.nv; base += NT) {       // 64 vertices per pass, the three lowest inserted in vertex order
    if (!((runs >> (base / NT)) & 1ull)) continue;
    const int vi = base + lane;
    const f4v q = h.vv[vi < h.nv ? vi : 0];
    const float lv[3] = {q.x, q.y, q.z};
    float wv[3];
    matvec(mR, lv, wv);
    for (int i = 0; i < 3; i++) wv[i] += mp[i];
    const float dist = (wv[0] - pp[0]) * n[0] + (wv[1] - pp[1]) * n[1] + (wv[2] - pp[2]) * n[2];
    unsigned long long neg = __ballot(vi < h.nv && dist < 0.f);
    while (neg) {
      const int v = __ffsll((long long)neg) - 1;
      neg &= neg - 1;
      const float dv = rl(dist, v), px = rl(wv[0], v), py = rl(wv[1], v), pz = rl(wv[2], v);
      int at = -1;
      if (nb < 1 || dv < best[0]) at = 0;
      else if (nb < 2 || dv < best[1]) at = 1;
      else if (nb < 3 || dv < best[2]) at = 2;
      if (at < 0) continue;
      if (at <= 1) { best[2] = best[1]; for (int i = 0; i < 3; i++) bpos[2][i] = bpos[1][i]; }
      if (at == 0) { best[1] = best[0]; for (int i = 0; i < 3; i++) bpos[1][i] = bpos[0][i]; }
      const int a_ = at;
      best[0] = a_ == 0 ? dv : best[0]; best[1] = a_ == 1 ? dv : best[1]; best[2] = a_ == 2 ? dv : best[2];
      bpos[0][0] = a_ == 0 ? px : bpos[0][0]; bpos[0][1] = a_ == 0 ? py : bpos[0][1]; bpos[0][2] = a_ == 0 ? pz : bpos[0][2];
      bpos[1][0] = a_ == 1 ? px : bpos[1][0]; bpos[1][1] = a_ == 1 ? py : bpos[1][1]; bpos[1][2] = a_ == 1 ? pz : bpos[1][2];
      bpos[2][0] = a_ == 2 ? px : bpos[2][0]; bpos[2][1] = a_ == 2 ? py : bpos[2][1]; bpos[2][2] = a_ == 2 ? pz : bpos[2][2];
      if (nb < 3) nb++;
    }
  }
  int cnt = 0;
#pragma unroll
  for (int s = 0; s < 3; s++) {
    if (s >= nb) continue;
    float pos[3];
    for (int i = 0; i < 3; i++) pos[i] = bpos[s][i] - 0.5f * best[s] * n[i];
    if (lane == 0) { LaneContacts o = owner; lc_put(o, cnt, best[s], pos, n); }
    cnt++;
  }
  return cnt;
}
__device__ __forceinline__ int col_capsule_mesh_wave(const HullRef& h, const float* cp, const float* cR, const float* cs,
                                                  const float* mp, const float* mR, const LaneContacts& owner) {
  float ax[3], rel[3], pc[3], al[3], a[3], d[3];
  matcol(cR, 2, ax);
  for (int i = 0; i < 3; i++) rel[i] = cp[i] - mp[i];
  mattvec(mR, rel, pc); mattvec(mR, ax, al);
  for (int i = 0; i < 3; i++) { a[i] = pc[i] - cs[1] * al[i]; d[i] = 2.f * cs[1] * al[i]; }
  const float r = cs[0];
  float p0[4], p1[4], pm[4], pl_[4], pr_[4];
  float v0, v1;
  HullSeg sg;
  {
    const float e1[3] = {a[0] + d[0], a[1] + d[1], a[2] + d[2]};
    hull_max_wave2(h, a, e1, p0, p1, v0, v1, sg);
  }
  const float s0 = p0[0] * d[0] + p0[1] * d[1] + p0[2] * d[2];
  const float s1 = p1[0] * d[0] + p1[1] * d[1] + p1[2] * d[2];
  float ts, vs, nmin[3];
  bool far = false;
  if (s0 >= 0.f) { ts = 0.f; vs = v0; for (int i = 0; i < 3; i++) nmin[i] = p0[i]; }
  else if (s1 <= 0.f) { ts = 1.f; vs = v1; for (int i = 0; i < 3; i++) nmin[i] = p1[i]; }
  else if (v0 < r && v1 < r) {      // both ends touch: the list is full with them, the interior minimum is never looked at
    ts = 0.f; vs = v0; for (int i = 0; i < 3; i++) nmin[i] = p0[i];
  } else {
    float tl = 0.f, vl = v0, sl = s0, tr = 1.f, vr = v1, sr = s1;
    for (int i = 0; i < 4; i++) { pl_[i] = p0[i]; pr_[i] = p1[i]; }
    ts = 0.f; vs = v0;
    for (int it = 0; it < 8; it++) {
      float t = fdiv((vr - sr * tr) - (vl - sl * tl), sl - sr);
      t = fminf(fmaxf(t, tl), tr);
      const float lineval = vl + sl * (t - tl);
      // the distance along the axis is convex and lies above both bracketing faces' lines, whose crossing is `lineval`: once
      // that is clear of the radius (by far more than the rounding of these few operations) no point of the axis is within
      // the radius, every value met so far included -- the remaining queries could only refine a minimum that is no contact
      if (lineval > r + 1e-6f) { far = true; break; }
      const float v = hull_max_wave_seg(h, a[0] + t * d[0], a[1] + t * d[1], a[2] + t * d[2], t, sg, pl_, pr_, pm);
      ts = t; vs = v;
      if (v <= lineval + 1e-7f) break;
      const float sm = pm[0] * d[0] + pm[1] * d[1] + pm[2] * d[2];
      if (sm < 0.f) { tl = t; vl = v; sl = sm; for (int i = 0; i < 4; i++) pl_[i] = pm[i]; }
      else { tr = t; vr = v; sr = sm; for (int i = 0; i < 4; i++) pr_[i] = pm[i]; }
    }
    const float lam = fdiv(sr, sr - sl);   // zero sub-gradient combination of the two tied faces
    for (int i = 0; i < 3; i++) nmin[i] = lam * pl_[i] + (1.f - lam) * pr_[i];
    normalize3(nmin);
  }
  // candidates in the order of the sequential list (end 0, end 1, interior minimum), at most two contacts
  const bool c0 = v0 < r, c1 = v1 < r;
  bool c2 = false;
  if (!(c0 && c1) && !far && vs < r) c2 = !((c0 && fabsf(ts) < 1e-4f) || (c1 && fabsf(ts - 1.f) < 1e-4f));
  int cnt = 0;
#pragma unroll
  for (int q = 0; q < 3; q++) {
    const bool on = q == 0 ? c0 : (q == 1 ? c1 : c2);
    if (!on || cnt >= 2) continue;
    const float* pl = q == 0 ? p0 : (q == 1 ? p1 : nmin);
    const float tq = q == 0 ? 0.f : (q == 1 ? 1.f : ts), vq = q == 0 ? v0 : (q == 1 ? v1 : vs);
    float pos[3], nrm[3], pw[3], nw[3];
    for (int i = 0; i < 3; i++) { const float c = a[i] + tq * d[i]; nrm[i] = -pl[i]; pos[i] = c - pl[i] * 0.5f * (r + vq); }
    matvec(mR, pos, pw); matvec(mR, nrm, nw);
    for (int i = 0; i < 3; i++) pw[i] += mp[i];
    if (threadIdx.x == 0) { LaneContacts o = owner; lc_put(o, cnt, vq - r, pw, nw); }
    cnt++;
  }
  return cnt;
}
__device__ __forceinline__ int col_box_mesh_wave(const HullRef& h, const float* bp, const float* bR, const float* bh,
                                              const float* mp, const float* mR, float mesh_rbound, const LaneContacts& owner) {
  const int lane = threadIdx.x;
  Deep4 keep{0, 0.f, 0.f, 0.f, 0.f};
  // runs of 64 vertices whose bounding sphere overlaps the box (lane = run)
  unsigned long long runs = ~0ull;
  if (h.prune) {
    bool reach = false;
    if (lane < h.nvr) {
      const f4v sp = h.vrun[lane];
      const float lc[3] = {sp.x, sp.y, sp.z};
      float wc[3], rel[3], pc[3];
      matvec(mR, lc, wc);
      for (int i = 0; i < 3; i++) rel[i] = wc[i] + mp[i] - bp[i];
      mattvec(bR, rel, pc);
      reach = true;
      for (int i = 0; i < 3; i++) if (fabsf(pc[i]) - sp.w > bh[i] + 1e-6f * (1.f + fabsf(pc[i]))) reach = false;
    }
    runs = __ballot(reach);
  }
  for (int base = 0; base < h.nv; base += NT) {   // hull vertices inside the box: 64 per pass, one per lane, kept in vertex order
    if (!((runs >> (base / NT)) & 1ull)) continue;
    const int vi = base + lane;
    const f4v q = h.vv[vi < h.nv ? vi : 0];
    const float lv[3] = {q.x, q.y, q.z};
    float wv[3], rel[3], p[3];
    matvec(mR, lv, wv);
    for (int i = 0; i < 3; i++) { wv[i] += mp[i]; rel[i] = wv[i] - bp[i]; }
    mattvec(bR, rel, p);
    float depth = 1e30f; int k = 0;
#pragma unroll
    for (int i = 0; i < 3; i++) { const float dd = bh[i] - fabsf(p[i]); if (dd < depth) { depth = dd; k = i; } }
    const float sg = (k == 0 ? p[0] : (k == 1 ? p[1] : p[2])) >= 0.f ? 1.f : -1.f;
    float nl[3] = {k == 0 ? sg : 0.f, k == 1 ? sg : 0.f, k == 2 ? sg : 0.f}, nw[3], pos[3];
    matvec(bR, nl, nw);
    for (int i = 0; i < 3; i++) pos[i] = wv[i] + nw[i] * 0.5f * depth;
    unsigned long long pen = __ballot(vi < h.nv && depth > 0.f);
    while (pen) {
      const int v = __ffsll((long long)pen) - 1;
      pen &= pen - 1;
      const float pv[3] = {rl(pos[0], v), rl(pos[1], v), rl(pos[2], v)}, nv[3] = {rl(nw[0], v), rl(nw[1], v), rl(nw[2], v)};
      deep4_add(keep, owner, -rl(depth, v), pv, nv);
    }
  }
  const float rb2 = mesh_rbound * mesh_rbound * 1.0001f + 1e-12f;
  for (int c = 0; c < 8; c++) {       // box corners inside the hull
    float loc[3] = {(c & 1 ? bh[0] : -bh[0]), (c & 2 ? bh[1] : -bh[1]), (c & 4 ? bh[2] : -bh[2])}, wc[3], rel[3], p[3];
    matvec(bR, loc, wc);
    for (int i = 0; i < 3; i++) { wc[i] += bp[i]; rel[i] = wc[i] - mp[i]; }
    mattvec(mR, rel, p);
    if (dot3(p, p) > rb2) continue;        // outside the hull's bounding sphere: cannot be inside the hull
    if (h.prune) {                         // outside the hull's bounding box (mesh frame = principal axes: tight for long shapes)
      bool out = false;
      for (int i = 0; i < 3; i++) { const float e = 1e-6f * (1.f + fabsf(p[i])); if (p[i] < h.lo[i] - e || p[i] > h.hi[i] + e) out = true; }
      if (out) continue;
    }
    float pf[4];
    const float s = hull_max_wave(h, p[0], p[1], p[2], pf);
    if (s >= 0.f) continue;
    float nl[3] = {-pf[0], -pf[1], -pf[2]}, nw[3], pos[3];
    matvec(mR, nl, nw);
    for (int i = 0; i < 3; i++) pos[i] = wc[i] + nw[i] * 0.5f * s;
    deep4_add(keep, owner, s, pos, nw);
  }
  return keep.n;
}

// tangents from the normal (same rule as the oracle's ho_make_frame)
HD void make_frame(float* f) {
  float* x = f; float* y = f + 3; float* z = f + 6;
  normalize3(x);
  if (fabsf(x[1]) < 0.5f) { y[0] = 0.f; y[1] = 1.f; y[2] = 0.f; } else { y[0] = 0.f; y[1] = 0.f; y[2] = 1.f; }
  float dp = dot3(x, y);
  for (int i = 0; i < 3; i++) y[i] -= dp * x[i];
  normalize3(y);
  cross3(x, y, z);
}

// Separating-axis test of two oriented boxes, all 15 axes: true = along one of them the boxes are more than `gap` apart, so
// nothing inside one is within `gap` of anything inside the other.  Box A: centre pa + Ra ca, axes = columns of Ra, half
// sizes a; box B: centre pb, axes = columns of Rb, half sizes b.  Conservative in rounding (a pair is only dropped with
// room to spare) and for nearly parallel edges (the usual epsilon on |R|).
HD bool obb_separated(const float* pa, const float* Ra, const float* ca, const float* a, const float* pb, const float* Rb, const float* b, float gap) {
  float R[3][3], Q[3][3], t[3];
  const float d[3] = {pb[0] - pa[0], pb[1] - pa[1], pb[2] - pa[2]};
#pragma unroll
  for (int i = 0; i < 3; i++) {
    t[i] = d[0] * Ra[i] + d[1] * Ra[3 + i] + d[2] * Ra[6 + i] - ca[i];
#pragma unroll
    for (int j = 0; j < 3; j++) { R[i][j] = Ra[i] * Rb[j] + Ra[3 + i] * Rb[3 + j] + Ra[6 + i] * Rb[6 + j]; Q[i][j] = fabsf(R[i][j]) + 1e-6f; }
  }
  const float g = gap + 1e-6f;
  bool sep = false;
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const float lhs = fabsf(t[i]);
    sep = sep || lhs > (a[i] + b[0] * Q[i][0] + b[1] * Q[i][1] + b[2] * Q[i][2] + g) * 1.00001f;
    const float lb = fabsf(t[0] * R[0][i] + t[1] * R[1][i] + t[2] * R[2][i]);
    sep = sep || lb > (b[i] + a[0] * Q[0][i] + a[1] * Q[1][i] + a[2] * Q[2][i] + g) * 1.00001f;
  }
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const int i1 = (i + 1) % 3, i2 = (i + 2) % 3;
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const int j1 = (j + 1) % 3, j2 = (j + 2) % 3;
      const float lhs = fabsf(t[i2] * R[i1][j] - t[i1] * R[i2][j]);
      sep = sep || lhs > (a[i1] * Q[i2][j] + a[i2] * Q[i1][j] + b[j1] * Q[i][j2] + b[j2] * Q[i][j1] + g) * 1.00001f;
    }
  }
  return sep;
}

// ---- collision driver: lane = pair (two passes when npair > 64); contacts compacted into the workspace
__device__ __forceinline__ void dev_collision(const DevModel& m, Work& w, int* overflow, int mesh_single) {
  const int tid = opaque(threadIdx.x);
  if (tid == 0) w.ncon = 0;
  wsync();
  for (int ps = 0; ps * NT < m.npair; ps++) {
    const int p = ps * NT + tid;
    const int pool = p < m.npair ? m.pair_pool[p] : -1;
    LaneContacts lc{0, &w.col_lc[tid], w.col_pool[max(pool, 0)], pool >= 0 ? COLSLOT + 2 : COLSLOT};
    int g1 = 0, g2 = 0;
    float pmargin = 0.f;
    bool isbb = false, ismesh = false;
    if (p < m.npair) {
      g1 = m.pair_geom1[p]; g2 = m.pair_geom2[p];
      const int t1 = m.pair_type1[p], t2 = m.pair_type2[p];
      const float bound = m.pair_bound[p];
      // (the pair's sizes and margin are fetched with its other constants, not behind the bounding-sphere test: one global-load
      //  latency for the stage instead of one per nesting level)
      const float z1[3] = {m.pair_size1[p][0], m.pair_size1[p][1], m.pair_size1[p][2]};
      const float z2[3] = {m.pair_size2[p][0], m.pair_size2[p][1], m.pair_size2[p][2]};
      pmargin = m.pair_margin[p];
      const float* p1 = w.gxpos[g1]; const float* R1 = w.gxmat[g1];
      const float* p2 = w.gxpos[g2]; const float* R2 = w.gxmat[g2];
      bool test;
      if (t1 != HOIC_GEOM_PLANE) {
        float dv[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]};
        test = dot3(dv, dv) <= bound * bound;
      } else {        // bounding sphere of geom2 against the plane (exact reject: no point of geom2 can be within the margin)
        test = (p2[0] - p1[0]) * R1[2] + (p2[1] - p1[1]) * R1[5] + (p2[2] - p1[2]) * R1[8] <= bound;
      }
      if (test && m.obb_reject && (t1 == HOIC_GEOM_CAPSULE || t1 == HOIC_GEOM_BOX) && (t2 == HOIC_GEOM_CAPSULE || t2 == HOIC_GEOM_BOX || t2 == HOIC_GEOM_MESH)) {
        // Second reject, one code path for every capsule / box / hull pair (no divergence between the pair types): both
        // geoms as oriented boxes -- a capsule is inside r x r x (l + r) along its axis, a hull inside its bounding box in
        // the mesh frame (the principal axes: tight for long objects) -- and the 15-axis separating-axis test with the
        // pair's margin (the oracle's driver makes the same test with the same constants, ho_sim.c ho_collision).  A pair it
        // drops is farther apart than the margin: the exact routines (capsule / box) return nothing for it anyway, and the
        // hull routines could only report what their max-over-face-planes distance under-estimates next to a sharp hull
        // vertex -- a shallow contact of two separated geoms that a geometric collider does not give (HOIC_NO_OBB_REJECT=1
        // switches the test off: tests/test_gpu_parity.py test_obb_reject_only_drops_contacts_of_separated_pairs).  What
        // it saves is the pair's turn in the sequential box-box / hull routines (four of five hull turns found nothing) and,
        // when no lane is left, the capsule-box routine.
        const bool cap1 = t1 == HOIC_GEOM_CAPSULE, cap2 = t2 == HOIC_GEOM_CAPSULE, hull2 = t2 == HOIC_GEOM_MESH;
        const float h1[3] = {z1[0], cap1 ? z1[0] : z1[1], cap1 ? z1[1] + z1[0] : z1[2]};
        float h2[3] = {z2[0], cap2 ? z2[0] : z2[1], cap2 ? z2[1] + z2[0] : z2[2]}, c2[3] = {0.f, 0.f, 0.f};
        if (hull2) {
          const int me = m.pair_mesh[p];
#pragma unroll
          for (int i = 0; i < 3; i++) { c2[i] = 0.5f * (m.mesh_aabb[me][i] + m.mesh_aabb[me][4 + i]); h2[i] = 0.5f * (m.mesh_aabb[me][4 + i] - m.mesh_aabb[me][i]); }
        }
        if (obb_separated(p2, R2, c2, h2, p1, R1, h1, pmargin)) test = false;
      }
      isbb = test && t1 == HOIC_GEOM_BOX && t2 == HOIC_GEOM_BOX;
      ismesh = test && t2 == HOIC_GEOM_MESH;
      if (ismesh && m.mesh_prune) {
        // second reject for mesh pairs: geom1's bounding sphere (a plane: the plane itself) against the hull's bounding BOX
        // in the mesh frame (the principal axes: a long object's box is far tighter than its bounding sphere).  A pair
        // farther apart than the margin can only produce contacts that the margin filter below drops: same contact list.
        const int me = m.pair_mesh[p];
        const float mg = pmargin;
        float q[3], d1[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]};
        mattvec(R2, d1, q);                              // geom1 centre (plane: a point of it) in the mesh frame
        if (t1 != HOIC_GEOM_PLANE) {
          const float r1 = bound - m.geom_rbound[g2];   // = rbound1 + margin
          float d2 = 0.f;
#pragma unroll
          for (int i = 0; i < 3; i++) { const float e = fmaxf(fmaxf(m.mesh_aabb[me][i] - q[i], q[i] - m.mesh_aabb[me][4 + i]), 0.f); d2 += e * e; }
          const float rr = r1 * (1.f + 1e-5f) + 1e-7f;
          if (d2 > rr * rr) ismesh = false;
        } else {
          float nl[3], pn[3] = {R1[2], R1[5], R1[8]};
          mattvec(R2, pn, nl);                           // plane normal in the mesh frame; q = a point of the plane
          float cdist = 0.f, ext = 0.f;
#pragma unroll
          for (int i = 0; i < 3; i++) {
            const float c = 0.5f * (m.mesh_aabb[me][i] + m.mesh_aabb[me][4 + i]), hh = 0.5f * (m.mesh_aabb[me][4 + i] - m.mesh_aabb[me][i]);
            cdist += (c - q[i]) * nl[i]; ext += hh * fabsf(nl[i]);
          }
          if (cdist - ext > mg + 1e-6f * (1.f + fabsf(cdist))) ismesh = false;      // the whole box is farther than the margin above the plane
        }
      }
      if (test && !isbb && !ismesh) {
        const float* s1 = z1; const float* s2 = z2;
        if (t1 == HOIC_GEOM_PLANE && t2 == HOIC_GEOM_CAPSULE) col_plane_capsule(p1, R1, p2, R2, s2, lc);
        else if (t1 == HOIC_GEOM_PLANE && t2 == HOIC_GEOM_BOX) col_plane_box(p1, R1, p2, R2, s2, lc);
        else if (t1 == HOIC_GEOM_CAPSULE && t2 == HOIC_GEOM_CAPSULE) col_capsule_capsule(p1, R1, s1, p2, R2, s2, lc);
        else if (t1 == HOIC_GEOM_CAPSULE && t2 == HOIC_GEOM_BOX) col_capsule_box(p1, R1, s1, p2, R2, s2, lc);
      }
    }
    // box-box pairs: one after the other, the whole wave on each (col_box_box_wave)
    {
      unsigned long long bbmask = __ballot(isbb);
#ifdef HOIC_TRACE_DISPATCH
      if (tid == 0) g_trace_bb[blockIdx.x] += __popcll(bbmask);
#endif
      while (bbmask) {
        const int L = __ffsll((long long)bbmask) - 1;
        bbmask &= bbmask - 1;
        const int pp = ps * NT + L, ga = m.pair_geom1[pp], gb = m.pair_geom2[pp];
        float Pa[3], RA[9], Ha[3], Pb[3], RB[9], Hb[3];
        for (int i = 0; i < 3; i++) { Pa[i] = w.gxpos[ga][i]; Pb[i] = w.gxpos[gb][i]; Ha[i] = m.pair_size1[pp][i]; Hb[i] = m.pair_size2[pp][i]; }
        for (int i = 0; i < 9; i++) { RA[i] = w.gxmat[ga][i]; RB[i] = w.gxmat[gb][i]; }
        const int pl_ = m.pair_pool[pp];
        const LaneContacts owner{0, &w.col_lc[L], w.col_pool[max(pl_, 0)], pl_ >= 0 ? COLSLOT + 2 : COLSLOT};
        const int nn = col_box_box_wave(Pa, RA, Ha, Pb, RB, Hb, owner, w.col_poly);
        if (tid == L) lc.n = min(nn, lc.cap);
      }
      wsync();
    }
    // mesh pairs: likewise one after the other, the hull tables streamed from the cache hierarchy
    {
      unsigned long long mm = __ballot(ismesh);
      while (mm) {
        const int L = __ffsll((long long)mm) - 1;
        mm &= mm - 1;
        const int pp = ps * NT + L, ga = m.pair_geom1[pp], gb = m.pair_geom2[pp], ta = m.pair_type1[pp], mesh = m.pair_mesh[pp];
        const HullRef hull = hull_ref(m, mesh);
        float Pa[3], RA[9], Sa[3], Pb[3], RB[9];
        for (int i = 0; i < 3; i++) { Pa[i] = w.gxpos[ga][i]; Pb[i] = w.gxpos[gb][i]; Sa[i] = m.pair_size1[pp][i]; }
        for (int i = 0; i < 9; i++) { RA[i] = w.gxmat[ga][i]; RB[i] = w.gxmat[gb][i]; }
        const int pl_ = m.pair_pool[pp];
        const LaneContacts owner{0, &w.col_lc[L], w.col_pool[max(pl_, 0)], pl_ >= 0 ? COLSLOT + 2 : COLSLOT};
        int nn;
        if (ta == HOIC_GEOM_CAPSULE) nn = col_capsule_mesh_wave(hull, Pa, RA, Sa, Pb, RB, owner);
        else if (ta == HOIC_GEOM_BOX) nn = col_box_mesh_wave(hull, Pa, RA, Sa, Pb, RB, m.geom_rbound[gb], owner);
        else nn = col_plane_mesh_wave(hull, Pa, RA, Pb, RB, owner);
        if (mesh_single && nn > 1) {     // hoic_env_config::mesh_single_contact: the deepest point only (the first one on ties)
          wsync();
          int bq = 0; float bd = lc_at(owner, 0, 0);
          for (int q = 1; q < nn; q++) { const float dq = lc_at(owner, q, 0); if (dq < bd) { bd = dq; bq = q; } }
          if (bq != 0 && tid < 7) { const float v = lc_at(owner, bq, tid); lc_at(owner, 0, tid) = v; }
          wsync();
          nn = 1;
        }
        if (tid == L) lc.n = min(nn, lc.cap);
      }
      wsync();
    }
    // margin filter, then the survivors go from the staging column to their positions in the contact list
    int cnt = 0;
    unsigned keepm = 0;
    const float margin = pmargin;
    if (p < m.npair) {
      for (int q = 0; q < lc.n; q++) if (lc_at(lc, q, 0) < margin) { keepm |= 1u << q; cnt++; }
    }
    // exclusive prefix sum of per-lane counts over the wave
    const int incl = wave_incl_scan(cnt);
    int c = w.ncon + incl - cnt;
    const int total = __builtin_amdgcn_readlane(incl, NT - 1);
    for (int q = 0; q < lc.n; q++) {
      if ((keepm >> q) & 1u) {
        if (c < MAXCON) {
          // (the frame is completed in registers and stored once: built in place in LDS, every step of make_frame was a
          //  dependent LDS round trip, four contacts of one pair one after the other)
          float fr[9], ps[3];
          const float dd = lc_at(lc, q, 0);
          for (int i = 0; i < 3; i++) { ps[i] = lc_at(lc, q, 1 + i); fr[i] = lc_at(lc, q, 4 + i); }
          make_frame(fr);
          w.c_dist[c] = dd; w.c_pair[c] = (unsigned char)p; w.c_g1[c] = (unsigned char)g1; w.c_g2[c] = (unsigned char)g2;
          for (int i = 0; i < 3; i++) w.c_pos[c][i] = ps[i];
          for (int i = 0; i < 9; i++) w.c_frame[c][i] = fr[i];
        }
        c++;
      }
    }
    wsync();
    if (tid == 0) {
      int nn = w.ncon + total;
      if (nn > MAXCON) { if (overflow) *overflow += 1; nn = MAXCON; }
      w.ncon = nn;
    }
    wsync();
  }
  // Constraint rows: 4 per condim-3 contact, 6 per condim-4 (object on the table), 1 per condim-1; the solver holds NCROW = 4
  // MAXCON of them.  The list is cut where the rows run out (a full list with condim-4 contacts in it): counted with the
  // contact overflows, never observed in a rollout (hoic_get_diagnostics).
  if (__builtin_amdgcn_readfirstlane(w.ncon) * 6 > NCROW) {      // (up to 21 contacts the rows fit whatever their condim)
    const int nc = w.ncon;
    int nr = 0;
    if (tid < nc) { const int dim = m.pair_condim[w.c_pair[tid]]; nr = dim == 1 ? 1 : 2 * (dim - 1); }
    const int incl = wave_incl_scan(nr);
    const int nfit = __popcll(__ballot(tid < nc && incl <= NCROW));
    if (nfit < nc) {
      if (tid == 0) { w.ncon = nfit; if (overflow) *overflow += 1; }
      wsync();
    }
  }
}
