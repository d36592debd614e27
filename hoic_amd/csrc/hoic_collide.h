// hoic_collide.h — narrow phase, one collision pair per lane.
//
// Replaces MuJoCo's collision stage for the static pair list of the HOIC models (the reference reads the
// result as data.contact[], uhc/envs/ho_im4.py:884-889).  Contacts follow MuJoCo's convention: dist < 0 on
// penetration, pos midway between the surfaces, normal from geom1 to geom2.  Same geometry as the CPU oracle
// (oracle/ho_collide.c) in float32; MuJoCo's exact contact multiset is not reproduced.
#pragma once
#include "hoic_types.h"
#include "hoic_math.h"

struct LaneContacts {
  int n;
  float dist[4], pos[4][3], nrm[4][3];
};

HD void lc_set(LaneContacts& o, int k, float dist, const float* pos, const float* n) {
  o.dist[k] = dist;
  for (int i = 0; i < 3; i++) { o.pos[k][i] = pos[i]; o.nrm[k][i] = n[i]; }
}

HD void col_plane_sphere(const float* pp, const float* pn, const float* c, float r, LaneContacts& o) {
  float d[3] = {c[0] - pp[0], c[1] - pp[1], c[2] - pp[2]};
  float dist = dot3(d, pn) - r;
  if (dist >= 0.f) return;
  float pos[3];
  for (int i = 0; i < 3; i++) pos[i] = c[i] - pn[i] * (r + 0.5f * dist);
  lc_set(o, o.n, dist, pos, pn);
  o.n++;
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
    lc_set(o, o.n, dist, pos, n);
    o.n++;
  }
}

HD void seg_seg_closest(const float* p1, const float* d1, const float* p2, const float* d2, float& s_out, float& t_out) {
  float r[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]};
  float a = dot3(d1, d1), e = dot3(d2, d2), f = dot3(d2, r), s, t;
  if (a <= 1e-12f && e <= 1e-12f) { s_out = t_out = 0.f; return; }
  if (a <= 1e-12f) { s = 0.f; t = fminf(fmaxf(f / e, 0.f), 1.f); }
  else {
    float c = dot3(d1, r);
    if (e <= 1e-12f) { t = 0.f; s = fminf(fmaxf(-c / a, 0.f), 1.f); }
    else {
      float b = dot3(d1, d2), den = a * e - b * b;
      s = den > 1e-6f * a * e ? fminf(fmaxf((b * f - c * e) / den, 0.f), 1.f) : 0.5f;
      t = (b * s + f) / e;
      if (t < 0.f) { t = 0.f; s = fminf(fmaxf(-c / a, 0.f), 1.f); }
      else if (t > 1.f) { t = 1.f; s = fminf(fmaxf((b - c) / a, 0.f), 1.f); }
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
  float len = sqrtf(dot3(d, d)), dist = len - s1[0] - s2[0];
  if (dist >= 0.f) return;
  if (len < 1e-12f) { d[0] = 1.f; d[1] = d[2] = 0.f; } else { float inv = 1.f / len; d[0] *= inv; d[1] *= inv; d[2] *= inv; }
  float pos[3];
  for (int i = 0; i < 3; i++) pos[i] = c1[i] + d[i] * (s1[0] + 0.5f * dist);
  lc_set(o, o.n, dist, pos, d);
  o.n++;
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
    float len = sqrtf(dot3(d, d));
    dist = len - r;
    if (dist >= 0.f) return false;
    float inv = 1.f / len;
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
    float tn = hh > 0.f ? t - g / hh : 0.5f * (lo + hi);
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
  float tc[3]; int nc = 0;
  float dist, pos[3], n[3];
  if (sphere_box_local(a, r, h, dist, pos, n)) tc[nc++] = 0.f;
  if (sphere_box_local(b, r, h, dist, pos, n)) tc[nc++] = 1.f;
  if (nc < 2) {
    float ts = seg_box_t(a, b, h);
    bool dup = false;
    for (int k = 0; k < nc; k++) if (fabsf(ts - tc[k]) < 1e-4f) dup = true;
    if (!dup) tc[nc++] = ts;
  }
  int cnt = 0;
  for (int k = 0; k < nc && cnt < 2; k++) {
    float c[3];
    for (int i = 0; i < 3; i++) c[i] = a[i] + tc[k] * (b[i] - a[i]);
    if (!sphere_box_local(c, r, h, dist, pos, n)) continue;
    float pw[3], nw[3];
    matvec(bR, pos, pw); matvec(bR, n, nw);
    for (int i = 0; i < 3; i++) pw[i] += bp[i];
    lc_set(o, o.n, dist, pw, nw);
    o.n++; cnt++;
  }
}

// ---- box-box: separating-axis test, then reference-face clipping (or one edge-edge point)
HD int clip_poly(const float (*p)[2], int n, int axis, float lim, float sgn, float (*q)[2]) {
  int mcount = 0;
  for (int i = 0; i < n; i++) {
    const float* a = p[i]; const float* b = p[(i + 1) % n];
    float da = sgn * a[axis] - lim, db = sgn * b[axis] - lim;
    if (da <= 0.f) { q[mcount][0] = a[0]; q[mcount][1] = a[1]; mcount++; }
    if ((da < 0.f && db > 0.f) || (da > 0.f && db < 0.f)) {
      float t = da / (da - db);
      q[mcount][0] = a[0] + t * (b[0] - a[0]); q[mcount][1] = a[1] + t * (b[1] - a[1]); mcount++;
    }
  }
  return mcount;
}
__device__ __forceinline__ void col_box_box(const float* pa, const float* Ra, const float* ha, const float* pb,
                                         const float* Rb, const float* hb, LaneContacts& o) {
  float A[3][3], B[3][3], R[3][3], Q[3][3], t[3], tw[3];
  for (int i = 0; i < 3; i++) { matcol(Ra, i, A[i]); matcol(Rb, i, B[i]); tw[i] = pb[i] - pa[i]; }
  for (int i = 0; i < 3; i++) {
    t[i] = dot3(tw, A[i]);
    for (int j = 0; j < 3; j++) { R[i][j] = dot3(A[i], B[j]); Q[i][j] = fabsf(R[i][j]) + 1e-6f; }
  }
  float best = 1e30f, bestn[3] = {0, 0, 0}; int code = -1;
  for (int i = 0; i < 3; i++) {
    float ra = ha[i], rb = hb[0] * Q[i][0] + hb[1] * Q[i][1] + hb[2] * Q[i][2];
    float pen = ra + rb - fabsf(t[i]);
    if (pen < 0.f) return;
    if (pen < best) { best = pen; code = i; float s = t[i] < 0.f ? -1.f : 1.f; for (int k = 0; k < 3; k++) bestn[k] = s * A[i][k]; }
  }
  for (int j = 0; j < 3; j++) {
    float tb = t[0] * R[0][j] + t[1] * R[1][j] + t[2] * R[2][j];
    float ra = ha[0] * Q[0][j] + ha[1] * Q[1][j] + ha[2] * Q[2][j], rb = hb[j];
    float pen = ra + rb - fabsf(tb);
    if (pen < 0.f) return;
    if (pen < best) { best = pen; code = 3 + j; float s = tb < 0.f ? -1.f : 1.f; for (int k = 0; k < 3; k++) bestn[k] = s * B[j][k]; }
  }
  float beste = 1e30f, en[3] = {0, 0, 0}; int ecode = -1;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      float L[3];
      cross3(A[i], B[j], L);
      float len = sqrtf(dot3(L, L));
      if (len < 1e-4f) continue;
      float inv = 1.f / len;
      for (int k = 0; k < 3; k++) L[k] *= inv;
      float ra = 0.f, rb = 0.f;
      for (int k = 0; k < 3; k++) { ra += ha[k] * fabsf(dot3(A[k], L)); rb += hb[k] * fabsf(dot3(B[k], L)); }
      float tl = dot3(tw, L), pen = ra + rb - fabsf(tl);
      if (pen < 0.f) return;
      if (pen < beste) { beste = pen; ecode = 3 * i + j; float s = tl < 0.f ? -1.f : 1.f; for (int k = 0; k < 3; k++) en[k] = s * L[k]; }
    }
  if (ecode >= 0 && beste * 1.05f + 1e-6f < best) {
    int i = ecode / 3, j = ecode % 3;
    float ea[3], eb[3];
    for (int k = 0; k < 3; k++) { ea[k] = pa[k]; eb[k] = pb[k]; }
    for (int k = 0; k < 3; k++) {
      if (k != i) { float s = dot3(en, A[k]) > 0.f ? 1.f : -1.f; for (int c = 0; c < 3; c++) ea[c] += s * ha[k] * A[k][c]; }
      if (k != j) { float s = dot3(en, B[k]) > 0.f ? -1.f : 1.f; for (int c = 0; c < 3; c++) eb[c] += s * hb[k] * B[k][c]; }
    }
    float r[3] = {ea[0] - eb[0], ea[1] - eb[1], ea[2] - eb[2]};
    float bdot = dot3(A[i], B[j]), c1 = dot3(A[i], r), f1 = dot3(B[j], r), den = 1.f - bdot * bdot;
    float u = den > 1e-6f ? (bdot * f1 - c1) / den : 0.f, v = f1 + bdot * u;
    u = fminf(fmaxf(u, -ha[i]), ha[i]); v = fminf(fmaxf(v, -hb[j]), hb[j]);
    float pos[3];
    for (int k = 0; k < 3; k++) pos[k] = 0.5f * (ea[k] + u * A[i][k] + eb[k] + v * B[j][k]);
    lc_set(o, 0, -beste, pos, en);
    o.n = 1;
    return;
  }
  const bool refA = code < 3; const int ax = refA ? code : code - 3;
  const float* pr = refA ? pa : pb; const float* pi_ = refA ? pb : pa;
  float(*Rr)[3] = refA ? A : B; float(*Ri)[3] = refA ? B : A;
  const float* hr = refA ? ha : hb; const float* hi = refA ? hb : ha;
  float nref[3];
  for (int k = 0; k < 3; k++) nref[k] = refA ? bestn[k] : -bestn[k];
  int iax = 0; float mind = 1e30f, isg = 1.f;
  for (int k = 0; k < 3; k++) {
    float dd = dot3(Ri[k], nref);
    if (-fabsf(dd) < mind) { mind = -fabsf(dd); iax = k; isg = dd > 0.f ? -1.f : 1.f; }
  }
  const int i1 = (iax + 1) % 3, i2 = (iax + 2) % 3, r1 = (ax + 1) % 3, r2 = (ax + 2) % 3;
  float fc[3];
  for (int k = 0; k < 3; k++) fc[k] = pi_[k] + isg * hi[iax] * Ri[iax][k] - pr[k];
  float poly[16][2], tmp[16][2], vz[4];
  const float sgs[4][2] = {{1, 1}, {-1, 1}, {-1, -1}, {1, -1}};
  for (int v = 0; v < 4; v++) {
    float wv[3];
    for (int k = 0; k < 3; k++) wv[k] = fc[k] + sgs[v][0] * hi[i1] * Ri[i1][k] + sgs[v][1] * hi[i2] * Ri[i2][k];
    poly[v][0] = dot3(wv, Rr[r1]); poly[v][1] = dot3(wv, Rr[r2]); vz[v] = dot3(wv, nref);
  }
  float m00 = poly[1][0] - poly[0][0], m01 = poly[1][1] - poly[0][1], m10 = poly[3][0] - poly[0][0], m11 = poly[3][1] - poly[0][1];
  float det = m00 * m11 - m01 * m10, gx = 0.f, gy = 0.f;
  if (fabsf(det) > 1e-12f) {
    float dz1 = vz[1] - vz[0], dz3 = vz[3] - vz[0];
    gx = (dz1 * m11 - dz3 * m01) / det; gy = (dz3 * m00 - dz1 * m10) / det;
  }
  float z0 = vz[0] - gx * poly[0][0] - gy * poly[0][1];
  int n = 4;
  n = clip_poly(poly, n, 0, hr[r1], 1.f, tmp); if (!n) return;
  n = clip_poly(tmp, n, 0, hr[r1], -1.f, poly); if (!n) return;
  n = clip_poly(poly, n, 1, hr[r2], 1.f, tmp); if (!n) return;
  n = clip_poly(tmp, n, 1, hr[r2], -1.f, poly); if (!n) return;
  float depth[16]; int keep[16], nk = 0;
  for (int v = 0; v < n; v++) {
    float z = z0 + gx * poly[v][0] + gy * poly[v][1];
    depth[v] = hr[ax] - z;
    if (depth[v] > 0.f) keep[nk++] = v;
  }
  if (!nk) return;
  int sel[4], ns = 0;
  if (nk <= 4) { for (int k = 0; k < nk; k++) sel[ns++] = keep[k]; }
  else {
    int d0 = 0;
    for (int k = 1; k < nk; k++) if (depth[keep[k]] > depth[keep[d0]]) d0 = k;
    for (int k = 0; k < 4; k++) sel[ns++] = keep[(d0 + (k * nk) / 4) % nk];
  }
  for (int s = 0; s < ns; s++) {
    int v = sel[s];
    float z = hr[ax] - depth[v], pos[3];
    for (int k = 0; k < 3; k++) pos[k] = pr[k] + poly[v][0] * Rr[r1][k] + poly[v][1] * Rr[r2][k] + (z + 0.5f * depth[v]) * nref[k];
    lc_set(o, s, -depth[v], pos, bestn);
  }
  o.n = ns;
}

// ---- convex mesh (hull vertices + face planes in the geom frame) vs plane / capsule / box.
// Plain loops over the hull tables, one pair per lane; see oracle/ho_collide.c for the formulation.
struct HullRef { const float (*v)[3]; int nv; const float (*pl)[4]; int np; };
HD HullRef get_hull(const DevModel& m, int mesh) {
  HullRef h;
  h.v = &m.mesh_vert[m.mesh_vertadr[mesh]]; h.nv = m.mesh_vertnum[mesh];
  h.pl = &m.mesh_plane[m.mesh_planeadr[mesh]]; h.np = m.mesh_planenum[mesh];
  return h;
}
HD float hull_line_max(const HullRef& h, const float* a, const float* d, float t, int& face) {
  const float x = a[0] + t * d[0], y = a[1] + t * d[1], z = a[2] + t * d[2];
  float best = -1e30f; int bf = 0;
  for (int f = 0; f < h.np; f++) {
    const float v = h.pl[f][0] * x + h.pl[f][1] * y + h.pl[f][2] * z - h.pl[f][3];
    if (v > best) { best = v; bf = f; }
  }
  face = bf;
  return best;
}
HD void lc_keep_deepest(LaneContacts& o, float dist, const float* pos, const float* n) {
  int slot = -1;
  if (o.n < 4) slot = o.n++;
  else {
    int wst = 0;
    for (int q = 1; q < 4; q++) if (o.dist[q] > o.dist[wst]) wst = q;
    if (dist < o.dist[wst]) slot = wst;
  }
  if (slot >= 0) {
    // slot is data dependent: write through selects so the arrays stay in registers
    for (int q = 0; q < 4; q++)
      if (q == slot) { o.dist[q] = dist; for (int i = 0; i < 3; i++) { o.pos[q][i] = pos[i]; o.nrm[q][i] = n[i]; } }
  }
}
__device__ __forceinline__ void col_plane_mesh(const DevModel& m, const float* pp, const float* pR, const float* mp,
                                            const float* mR, int mesh, LaneContacts& o) {
  const HullRef h = get_hull(m, mesh);
  float n[3];
  matcol(pR, 2, n);
  float best[3] = {0.f, 0.f, 0.f}; int bi[3] = {-1, -1, -1};
  for (int v = 0; v < h.nv; v++) {
    float wv[3];
    matvec(mR, h.v[v], wv);
    const float dist = (wv[0] + mp[0] - pp[0]) * n[0] + (wv[1] + mp[1] - pp[1]) * n[1] + (wv[2] + mp[2] - pp[2]) * n[2];
    if (dist >= 0.f) continue;
    if (bi[0] < 0 || dist < best[0]) { best[2] = best[1]; bi[2] = bi[1]; best[1] = best[0]; bi[1] = bi[0]; best[0] = dist; bi[0] = v; }
    else if (bi[1] < 0 || dist < best[1]) { best[2] = best[1]; bi[2] = bi[1]; best[1] = dist; bi[1] = v; }
    else if (bi[2] < 0 || dist < best[2]) { best[2] = dist; bi[2] = v; }
  }
  for (int s = 0; s < 3; s++) {
    if (bi[s] < 0) continue;
    float wv[3], pos[3];
    matvec(mR, h.v[bi[s]], wv);
    for (int i = 0; i < 3; i++) pos[i] = wv[i] + mp[i] - 0.5f * best[s] * n[i];
    lc_set(o, o.n, best[s], pos, n);
    o.n++;
  }
}
__device__ __forceinline__ void col_capsule_mesh(const DevModel& m, const float* cp, const float* cR, const float* cs,
                                              const float* mp, const float* mR, int mesh, LaneContacts& o) {
  const HullRef h = get_hull(m, mesh);
  float ax[3], rel[3], pc[3], al[3], a[3], d[3];
  matcol(cR, 2, ax);
  for (int i = 0; i < 3; i++) rel[i] = cp[i] - mp[i];
  mattvec(mR, rel, pc); mattvec(mR, ax, al);
  for (int i = 0; i < 3; i++) { a[i] = pc[i] - cs[1] * al[i]; d[i] = 2.f * cs[1] * al[i]; }
  const float r = cs[0];
  int f0, f1, fm;
  const float v0 = hull_line_max(h, a, d, 0.f, f0), v1 = hull_line_max(h, a, d, 1.f, f1);
  const float s0 = h.pl[f0][0] * d[0] + h.pl[f0][1] * d[1] + h.pl[f0][2] * d[2];
  const float s1 = h.pl[f1][0] * d[0] + h.pl[f1][1] * d[1] + h.pl[f1][2] * d[2];
  float ts, vs, nmin[3];
  if (s0 >= 0.f) { ts = 0.f; vs = v0; for (int i = 0; i < 3; i++) nmin[i] = h.pl[f0][i]; }
  else if (s1 <= 0.f) { ts = 1.f; vs = v1; for (int i = 0; i < 3; i++) nmin[i] = h.pl[f1][i]; }
  else {
    float tl = 0.f, vl = v0, sl = s0, tr = 1.f, vr = v1, sr = s1; int fl = f0, fr = f1;
    ts = 0.f; vs = v0;
    for (int it = 0; it < 8; it++) {
      float t = ((vr - sr * tr) - (vl - sl * tl)) / (sl - sr);
      t = fminf(fmaxf(t, tl), tr);
      const float v = hull_line_max(h, a, d, t, fm);
      const float lineval = vl + sl * (t - tl);
      ts = t; vs = v;
      if (v <= lineval + 1e-7f) break;
      const float sm = h.pl[fm][0] * d[0] + h.pl[fm][1] * d[1] + h.pl[fm][2] * d[2];
      if (sm < 0.f) { tl = t; vl = v; sl = sm; fl = fm; } else { tr = t; vr = v; sr = sm; fr = fm; }
    }
    const float lam = sr / (sr - sl);   // zero sub-gradient combination of the two tied faces
    for (int i = 0; i < 3; i++) nmin[i] = lam * h.pl[fl][i] + (1.f - lam) * h.pl[fr][i];
    normalize3(nmin);
  }
  float tc[3], vc[3], nc3[3][3]; int nc = 0;
  if (v0 < r) { tc[nc] = 0.f; vc[nc] = v0; for (int i = 0; i < 3; i++) nc3[nc][i] = h.pl[f0][i]; nc++; }
  if (v1 < r) { tc[nc] = 1.f; vc[nc] = v1; for (int i = 0; i < 3; i++) nc3[nc][i] = h.pl[f1][i]; nc++; }
  if (nc < 2 && vs < r) {
    bool dup = false;
    for (int q = 0; q < nc; q++) if (fabsf(ts - tc[q]) < 1e-4f) dup = true;
    if (!dup) { tc[nc] = ts; vc[nc] = vs; for (int i = 0; i < 3; i++) nc3[nc][i] = nmin[i]; nc++; }
  }
  for (int q = 0; q < nc && q < 2; q++) {
    const float* pl = nc3[q];
    float pos[3], nrm[3], pw[3], nw[3];
    for (int i = 0; i < 3; i++) { const float c = a[i] + tc[q] * d[i]; nrm[i] = -pl[i]; pos[i] = c - pl[i] * 0.5f * (r + vc[q]); }
    matvec(mR, pos, pw); matvec(mR, nrm, nw);
    for (int i = 0; i < 3; i++) pw[i] += mp[i];
    lc_set(o, o.n, vc[q] - r, pw, nw);
    o.n++;
  }
}
__device__ __forceinline__ void col_box_mesh(const DevModel& m, const float* bp, const float* bR, const float* bh,
                                          const float* mp, const float* mR, int mesh, LaneContacts& o) {
  const HullRef h = get_hull(m, mesh);
  for (int v = 0; v < h.nv; v++) {
    float wv[3], rel[3], p[3];
    matvec(mR, h.v[v], wv);
    for (int i = 0; i < 3; i++) { wv[i] += mp[i]; rel[i] = wv[i] - bp[i]; }
    mattvec(bR, rel, p);
    float depth = 1e30f; int k = 0;
    for (int i = 0; i < 3; i++) { const float dd = bh[i] - fabsf(p[i]); if (dd < depth) { depth = dd; k = i; } }
    if (depth <= 0.f) continue;
    float nl[3] = {0.f, 0.f, 0.f}, nw[3], pos[3];
    const float sg = p[k] >= 0.f ? 1.f : -1.f;
    nl[0] = k == 0 ? sg : 0.f; nl[1] = k == 1 ? sg : 0.f; nl[2] = k == 2 ? sg : 0.f;
    matvec(bR, nl, nw);
    for (int i = 0; i < 3; i++) pos[i] = wv[i] + nw[i] * 0.5f * depth;
    lc_keep_deepest(o, -depth, pos, nw);
  }
  for (int c = 0; c < 8; c++) {
    float loc[3] = {(c & 1 ? bh[0] : -bh[0]), (c & 2 ? bh[1] : -bh[1]), (c & 4 ? bh[2] : -bh[2])}, wc[3], rel[3], p[3];
    matvec(bR, loc, wc);
    for (int i = 0; i < 3; i++) { wc[i] += bp[i]; rel[i] = wc[i] - mp[i]; }
    mattvec(mR, rel, p);
    const float zero[3] = {0.f, 0.f, 0.f}; int f;
    const float s = hull_line_max(h, p, zero, 0.f, f);
    if (s >= 0.f) continue;
    float nl[3] = {-h.pl[f][0], -h.pl[f][1], -h.pl[f][2]}, nw[3], pos[3];
    matvec(mR, nl, nw);
    for (int i = 0; i < 3; i++) pos[i] = wc[i] + nw[i] * 0.5f * s;
    lc_keep_deepest(o, s, pos, nw);
  }
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

// ---- collision driver: lane = pair (two passes when npair > 64); contacts compacted into the workspace
__device__ __forceinline__ void dev_collision(const DevModel& m, Work& w, int* overflow) {
  const int tid = threadIdx.x;
  if (tid == 0) w.ncon = 0;
  __syncthreads();
  for (int ps = 0; ps * NT < m.npair; ps++) {
    const int p = ps * NT + tid;
    LaneContacts lc;
    lc.n = 0;
    int g1 = 0, g2 = 0;
    if (p < m.npair) {
      g1 = m.pair_geom1[p]; g2 = m.pair_geom2[p];
      const int t1 = m.pair_type1[p], t2 = m.pair_type2[p];
      const float bound = m.pair_bound[p], margin = m.pair_margin[p];
      const float* p1 = w.gxpos[g1]; const float* R1 = w.gxmat[g1];
      const float* p2 = w.gxpos[g2]; const float* R2 = w.gxmat[g2];
      bool test = true;
      if (t1 != HOIC_GEOM_PLANE) {
        float dv[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]};
        test = dot3(dv, dv) <= bound * bound;
      }
      if (test) {
        const float s1[3] = {m.geom_size[g1][0], m.geom_size[g1][1], m.geom_size[g1][2]};
        const float s2[3] = {m.geom_size[g2][0], m.geom_size[g2][1], m.geom_size[g2][2]};
        if (t1 == HOIC_GEOM_PLANE && t2 == HOIC_GEOM_CAPSULE) col_plane_capsule(p1, R1, p2, R2, s2, lc);
        else if (t1 == HOIC_GEOM_PLANE && t2 == HOIC_GEOM_BOX) col_plane_box(p1, R1, p2, R2, s2, lc);
        else if (t1 == HOIC_GEOM_CAPSULE && t2 == HOIC_GEOM_CAPSULE) col_capsule_capsule(p1, R1, s1, p2, R2, s2, lc);
        else if (t1 == HOIC_GEOM_CAPSULE && t2 == HOIC_GEOM_BOX) col_capsule_box(p1, R1, s1, p2, R2, s2, lc);
        else if (t1 == HOIC_GEOM_BOX && t2 == HOIC_GEOM_BOX) col_box_box(p1, R1, s1, p2, R2, s2, lc);
        else if (t2 == HOIC_GEOM_MESH) {
          const int mesh = m.pair_mesh[p];
          if (t1 == HOIC_GEOM_CAPSULE) col_capsule_mesh(m, p1, R1, s1, p2, R2, mesh, lc);
          else if (t1 == HOIC_GEOM_BOX) col_box_mesh(m, p1, R1, s1, p2, R2, mesh, lc);
          else if (t1 == HOIC_GEOM_PLANE) col_plane_mesh(m, p1, R1, p2, R2, mesh, lc);
        }
      }
      // margin filter
      int k2 = 0;
      for (int q = 0; q < lc.n; q++)
        if (lc.dist[q] < margin) {
          if (k2 != q) { lc.dist[k2] = lc.dist[q]; for (int i = 0; i < 3; i++) { lc.pos[k2][i] = lc.pos[q][i]; lc.nrm[k2][i] = lc.nrm[q][i]; } }
          k2++;
        }
      lc.n = k2;
    }
    // exclusive prefix sum of per-lane counts over the wave
    const int cnt = lc.n, incl = wave_incl_scan(cnt);
    const int start = w.ncon + incl - cnt;
    const int total = __builtin_amdgcn_readlane(incl, NT - 1);
    for (int q = 0; q < lc.n; q++) {
      const int c = start + q;
      if (c < MAXCON) {
        w.c_dist[c] = lc.dist[q]; w.c_pair[c] = (unsigned char)p; w.c_g1[c] = (unsigned char)g1; w.c_g2[c] = (unsigned char)g2;
        for (int i = 0; i < 3; i++) { w.c_pos[c][i] = lc.pos[q][i]; w.c_frame[c][i] = lc.nrm[q][i]; }
        make_frame(w.c_frame[c]);
      }
    }
    __syncthreads();
    if (tid == 0) {
      int nn = w.ncon + total;
      if (nn > MAXCON) { if (overflow) *overflow += 1; nn = MAXCON; }
      w.ncon = nn;
    }
    __syncthreads();
  }
}
