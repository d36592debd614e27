/* ho_collide.c — CPU oracle, narrow phase (TEST INFRASTRUCTURE).
 *
 * The reference gets contacts from MuJoCo's collision functions (data.contact, read at
 * uhc/envs/ho_im4.py:884-889).  MuJoCo is not in /root/reference, so these are from-scratch
 * geometric routines producing MuJoCo-CONVENTION contacts [MJ-doc]: dist < 0 on penetration,
 * pos = midpoint of the two surface points, frame[0:3] = normal pointing from geom1 to geom2.
 * The contact multiset of MuJoCo's own routines (mjc_BoxBox, mjc_CapsuleBox, libccd for meshes) is
 * NOT reproduced point-for-point: PARITY UNPINNED.
 */
#include "ho_oracle.h"
#include <math.h>
#include <string.h>

static void col(const double R[9], int k, double o[3]) { o[0] = R[k]; o[1] = R[3 + k]; o[2] = R[6 + k]; }
static void mtv(const double R[9], const double v[3], double o[3]) { /* R^T v */
  double x = R[0] * v[0] + R[3] * v[1] + R[6] * v[2], y = R[1] * v[0] + R[4] * v[1] + R[7] * v[2],
         z = R[2] * v[0] + R[5] * v[1] + R[8] * v[2];
  o[0] = x; o[1] = y; o[2] = z;
}
static void mv(const double R[9], const double v[3], double o[3]) {
  double x = R[0] * v[0] + R[1] * v[1] + R[2] * v[2], y = R[3] * v[0] + R[4] * v[1] + R[5] * v[2],
         z = R[6] * v[0] + R[7] * v[1] + R[8] * v[2];
  o[0] = x; o[1] = y; o[2] = z;
}
static void set_contact(ho_contact* c, double dist, const double pos[3], const double n[3]) {
  memset(c, 0, sizeof(*c));
  c->dist = dist;
  for (int i = 0; i < 3; i++) { c->pos[i] = pos[i]; c->frame[i] = n[i]; }
}

/* ---------------------------------------------------------------- plane vs sphere-swept shapes */
static int plane_sphere(const double pp[3], const double pn[3], const double c[3], double r, ho_contact* out) {
  double d[3] = {c[0] - pp[0], c[1] - pp[1], c[2] - pp[2]};
  double dist = ho_dot3(d, pn) - r;
  if (dist >= 0) return 0;
  double pos[3];
  for (int i = 0; i < 3; i++) pos[i] = c[i] - pn[i] * (r + 0.5 * dist);
  set_contact(out, dist, pos, pn);
  return 1;
}
static int plane_capsule(const double pp[3], const double pR[9], const double cp[3], const double cR[9],
                         const double size[3], ho_contact* out) {
  double n[3], ax[3], e[3];
  int cnt = 0;
  col(pR, 2, n); col(cR, 2, ax);
  for (int s = -1; s <= 1; s += 2) {
    for (int i = 0; i < 3; i++) e[i] = cp[i] + s * size[1] * ax[i];
    cnt += plane_sphere(pp, n, e, size[0], out + cnt);
  }
  return cnt;
}
static int plane_box(const double pp[3], const double pR[9], const double bp[3], const double bR[9],
                     const double h[3], ho_contact* out) {
  double n[3];
  int cnt = 0;
  col(pR, 2, n);
  for (int k = 0; k < 8 && cnt < 4; k++) {
    double loc[3] = {(k & 1 ? h[0] : -h[0]), (k & 2 ? h[1] : -h[1]), (k & 4 ? h[2] : -h[2])}, w[3], d[3];
    mv(bR, loc, w);
    for (int i = 0; i < 3; i++) { w[i] += bp[i]; d[i] = w[i] - pp[i]; }
    double dist = ho_dot3(d, n);
    if (dist >= 0) continue;
    double pos[3];
    for (int i = 0; i < 3; i++) pos[i] = w[i] - 0.5 * dist * n[i];
    set_contact(out + cnt, dist, pos, n);
    cnt++;
  }
  return cnt;
}
static int plane_mesh(const ho_model* m, const double pp[3], const double pR[9], const double mp[3],
                      const double mR[9], int mesh, ho_contact* out) {
  /* up to 3 deepest hull vertices below the plane */
  double n[3];
  col(pR, 2, n);
  int cnt = 0;
  double best[3] = {0, 0, 0}; int bi[3] = {-1, -1, -1};
  for (int v = 0; v < m->mesh_vertnum[mesh]; v++) {
    double w[3], d[3];
    mv(mR, m->mesh_vert[m->mesh_vertadr[mesh] + v], w);
    for (int i = 0; i < 3; i++) { w[i] += mp[i]; d[i] = w[i] - pp[i]; }
    double dist = ho_dot3(d, n);
    if (dist >= 0) continue;
    for (int s = 0; s < 3; s++)
      if (bi[s] < 0 || dist < best[s]) {
        for (int t = 2; t > s; t--) { best[t] = best[t - 1]; bi[t] = bi[t - 1]; }
        best[s] = dist; bi[s] = v; break;
      }
  }
  for (int s = 0; s < 3; s++) {
    if (bi[s] < 0) continue;
    double w[3], pos[3];
    mv(mR, m->mesh_vert[m->mesh_vertadr[mesh] + bi[s]], w);
    for (int i = 0; i < 3; i++) pos[i] = w[i] + mp[i] - 0.5 * best[s] * n[i];
    set_contact(out + cnt, best[s], pos, n);
    cnt++;
  }
  return cnt;
}

/* ---------------------------------------------------------------- capsule - capsule */
static void seg_seg_closest(const double p1[3], const double d1[3], const double p2[3], const double d2[3],
                            double* s_out, double* t_out) {
  /* segments p1 + s d1, p2 + t d2, s,t in [0,1] */
  double r[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]};
  double a = ho_dot3(d1, d1), e = ho_dot3(d2, d2), f = ho_dot3(d2, r), s, t;
  if (a <= HO_MINVAL && e <= HO_MINVAL) { *s_out = *t_out = 0; return; }
  if (a <= HO_MINVAL) { s = 0; t = fmin(fmax(f / e, 0), 1); }
  else {
    double c = ho_dot3(d1, r);
    if (e <= HO_MINVAL) { t = 0; s = fmin(fmax(-c / a, 0), 1); }
    else {
      double b = ho_dot3(d1, d2), den = a * e - b * b;
      s = den > 1e-12 * a * e ? fmin(fmax((b * f - c * e) / den, 0), 1) : 0.5; /* parallel: midpoint */
      t = (b * s + f) / e;
      if (t < 0) { t = 0; s = fmin(fmax(-c / a, 0), 1); }
      else if (t > 1) { t = 1; s = fmin(fmax((b - c) / a, 0), 1); }
    }
  }
  *s_out = s; *t_out = t;
}
static int sphere_sphere(const double c1[3], double r1, const double c2[3], double r2, ho_contact* out) {
  double d[3] = {c2[0] - c1[0], c2[1] - c1[1], c2[2] - c1[2]};
  double len = sqrt(ho_dot3(d, d)), dist = len - r1 - r2;
  if (dist >= 0) return 0;
  if (len < HO_MINVAL) { d[0] = 1; d[1] = d[2] = 0; } else { d[0] /= len; d[1] /= len; d[2] /= len; }
  double pos[3];
  for (int i = 0; i < 3; i++) pos[i] = c1[i] + d[i] * (r1 + 0.5 * dist);
  set_contact(out, dist, pos, d);
  return 1;
}
static int capsule_capsule(const double p1[3], const double R1[9], const double s1[3], const double p2[3],
                           const double R2[9], const double s2[3], ho_contact* out) {
  double a1[3], a2[3], q1[3], q2[3], d1[3], d2[3], s, t, c1[3], c2[3];
  col(R1, 2, a1); col(R2, 2, a2);
  for (int i = 0; i < 3; i++) {
    q1[i] = p1[i] - s1[1] * a1[i]; d1[i] = 2 * s1[1] * a1[i];
    q2[i] = p2[i] - s2[1] * a2[i]; d2[i] = 2 * s2[1] * a2[i];
  }
  seg_seg_closest(q1, d1, q2, d2, &s, &t);
  for (int i = 0; i < 3; i++) { c1[i] = q1[i] + s * d1[i]; c2[i] = q2[i] + t * d2[i]; }
  return sphere_sphere(c1, s1[0], c2, s2[0], out);
}

/* ---------------------------------------------------------------- capsule - box */
/* sphere (centre c in the box frame) against the box; result in box frame */
static int sphere_box_local(const double c[3], double r, const double h[3], double* dist, double pos[3], double n[3]) {
  double q[3], d[3];
  int inside = 1;
  for (int i = 0; i < 3; i++) {
    q[i] = fmin(fmax(c[i], -h[i]), h[i]);
    d[i] = c[i] - q[i];
    if (d[i] != 0) inside = 0;
  }
  if (!inside) {
    double len = sqrt(ho_dot3(d, d));
    *dist = len - r;
    if (*dist >= 0) return 0;
    for (int i = 0; i < 3; i++) { n[i] = -d[i] / len; pos[i] = q[i] + d[i] / len * 0.5 * (*dist); }
    return 1;
  }
  /* centre inside the box: push out through the nearest face */
  int k = 0; double best = 1e300;
  for (int i = 0; i < 3; i++) { double dep = h[i] - fabs(c[i]); if (dep < best) { best = dep; k = i; } }
  double sg = c[k] >= 0 ? 1.0 : -1.0;
  *dist = -(best + r);
  for (int i = 0; i < 3; i++) { n[i] = 0; pos[i] = c[i]; }
  n[k] = -sg;
  pos[k] = c[k] + sg * 0.5 * (best - r);
  return 1;
}
/* squared distance from segment point a + t*(b-a) to the box and its derivative pieces */
static double seg_box_t(const double a[3], const double b[3], const double h[3]) {
  /* minimise the convex piecewise-quadratic f(t) = dist^2(P(t), box) over [0,1] */
  double t = 0.5, lo = 0, hi = 1;
  for (int it = 0; it < 60; it++) {
    double g = 0, hh = 0;
    for (int i = 0; i < 3; i++) {
      double p = a[i] + t * (b[i] - a[i]), v = b[i] - a[i];
      double ex = p > h[i] ? p - h[i] : (p < -h[i] ? p + h[i] : 0);
      if (ex != 0) { g += 2 * ex * v; hh += 2 * v * v; }
    }
    if (g > 0) hi = t; else if (g < 0) lo = t; else break;
    double tn = hh > 0 ? t - g / hh : 0.5 * (lo + hi);
    if (tn <= lo || tn >= hi) tn = 0.5 * (lo + hi);
    if (fabs(tn - t) < 1e-15) { t = tn; break; }
    t = tn;
    if (hi - lo < 1e-15) break;
  }
  return t;
}
static int capsule_box(const double cp[3], const double cR[9], const double cs[3], const double bp[3],
                       const double bR[9], const double h[3], ho_contact* out) {
  double ax[3], rel[3], pc[3], al[3], a[3], b[3];
  col(cR, 2, ax);
  for (int i = 0; i < 3; i++) rel[i] = cp[i] - bp[i];
  mtv(bR, rel, pc); mtv(bR, ax, al);
  for (int i = 0; i < 3; i++) { a[i] = pc[i] - cs[1] * al[i]; b[i] = pc[i] + cs[1] * al[i]; }
  double r = cs[0];
  double tcand[3]; int nc = 0;
  /* penetrating end spheres first, then the closest point if distinct */
  double dist, pos[3], n[3];
  int endhit[2] = {0, 0};
  if (sphere_box_local(a, r, h, &dist, pos, n)) { endhit[0] = 1; tcand[nc++] = 0; }
  if (sphere_box_local(b, r, h, &dist, pos, n)) { endhit[1] = 1; tcand[nc++] = 1; }
  if (nc < 2) {
    double ts = seg_box_t(a, b, h);
    int dup = 0;
    for (int k = 0; k < nc; k++) if (fabs(ts - tcand[k]) < 1e-6) dup = 1;
    if (!dup) tcand[nc++] = ts;
  }
  int cnt = 0;
  for (int k = 0; k < nc && cnt < 2; k++) {
    double c[3];
    for (int i = 0; i < 3; i++) c[i] = a[i] + tcand[k] * (b[i] - a[i]);
    if (!sphere_box_local(c, r, h, &dist, pos, n)) continue;
    double pw[3], nw[3];
    mv(bR, pos, pw); mv(bR, n, nw);
    for (int i = 0; i < 3; i++) pw[i] += bp[i];
    set_contact(out + cnt, dist, pw, nw);
    cnt++;
  }
  (void)endhit;
  return cnt;
}

/* ---------------------------------------------------------------- box - box (SAT + face clipping) */
static int clip_poly(double (*p)[2], int n, int axis, double lim, double sgn, double (*q)[2]) {
  /* keep sgn * p[axis] <= lim */
  int m = 0;
  for (int i = 0; i < n; i++) {
    double* a = p[i]; double* b = p[(i + 1) % n];
    double da = sgn * a[axis] - lim, db = sgn * b[axis] - lim;
    if (da <= 0) { q[m][0] = a[0]; q[m][1] = a[1]; m++; }
    if ((da < 0 && db > 0) || (da > 0 && db < 0)) {
      double t = da / (da - db);
      q[m][0] = a[0] + t * (b[0] - a[0]); q[m][1] = a[1] + t * (b[1] - a[1]); m++;
    }
  }
  return m;
}
static int box_box(const double pa[3], const double Ra[9], const double ha[3], const double pb[3],
                   const double Rb[9], const double hb[3], ho_contact* out, int maxout) {
  double A[3][3], B[3][3], R[3][3], Q[3][3], t[3], tw[3];
  for (int i = 0; i < 3; i++) { col(Ra, i, A[i]); col(Rb, i, B[i]); tw[i] = pb[i] - pa[i]; }
  for (int i = 0; i < 3; i++) {
    t[i] = ho_dot3(tw, A[i]);
    for (int j = 0; j < 3; j++) { R[i][j] = ho_dot3(A[i], B[j]); Q[i][j] = fabs(R[i][j]) + 1e-12; }
  }
  double best = 1e300, bestn[3] = {0, 0, 0}; int code = -1;
  /* face axes of A */
  for (int i = 0; i < 3; i++) {
    double ra = ha[i], rb = hb[0] * Q[i][0] + hb[1] * Q[i][1] + hb[2] * Q[i][2];
    double pen = ra + rb - fabs(t[i]);
    if (pen < 0) return 0;
    if (pen < best) { best = pen; code = i; double s = t[i] < 0 ? -1 : 1; for (int k = 0; k < 3; k++) bestn[k] = s * A[i][k]; }
  }
  /* face axes of B */
  for (int j = 0; j < 3; j++) {
    double tb = t[0] * R[0][j] + t[1] * R[1][j] + t[2] * R[2][j];
    double ra = ha[0] * Q[0][j] + ha[1] * Q[1][j] + ha[2] * Q[2][j], rb = hb[j];
    double pen = ra + rb - fabs(tb);
    if (pen < 0) return 0;
    if (pen < best) { best = pen; code = 3 + j; double s = tb < 0 ? -1 : 1; for (int k = 0; k < 3; k++) bestn[k] = s * B[j][k]; }
  }
  /* edge x edge axes; chosen only if clearly better than the best face axis */
  double beste = 1e300, en[3] = {0, 0, 0}; int ecode = -1;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double L[3];
      ho_cross(A[i], B[j], L);
      double len = sqrt(ho_dot3(L, L));
      if (len < 1e-6) continue;
      for (int k = 0; k < 3; k++) L[k] /= len;
      double ra = 0, rb = 0;
      for (int k = 0; k < 3; k++) { ra += ha[k] * fabs(ho_dot3(A[k], L)); rb += hb[k] * fabs(ho_dot3(B[k], L)); }
      double tl = ho_dot3(tw, L), pen = ra + rb - fabs(tl);
      if (pen < 0) return 0;
      if (pen < beste) { beste = pen; ecode = 3 * i + j; double s = tl < 0 ? -1 : 1; for (int k = 0; k < 3; k++) en[k] = s * L[k]; }
    }
  if (ecode >= 0 && beste * 1.05 + 1e-9 < best) {
    /* edge-edge contact: supporting edges, closest points of the two lines */
    int i = ecode / 3, j = ecode % 3;
    double ea[3], eb[3];
    for (int k = 0; k < 3; k++) { ea[k] = pa[k]; eb[k] = pb[k]; }
    for (int k = 0; k < 3; k++) {
      if (k != i) { double s = ho_dot3(en, A[k]) > 0 ? 1 : -1; for (int c = 0; c < 3; c++) ea[c] += s * ha[k] * A[k][c]; }
      if (k != j) { double s = ho_dot3(en, B[k]) > 0 ? -1 : 1; for (int c = 0; c < 3; c++) eb[c] += s * hb[k] * B[k][c]; }
    }
    /* lines ea + u A[i], eb + v B[j] */
    double r[3] = {ea[0] - eb[0], ea[1] - eb[1], ea[2] - eb[2]};
    double bdot = ho_dot3(A[i], B[j]), c1 = ho_dot3(A[i], r), f1 = ho_dot3(B[j], r), den = 1 - bdot * bdot;
    double u = den > 1e-12 ? (bdot * f1 - c1) / den : 0, v = f1 + bdot * u;
    u = fmin(fmax(u, -ha[i]), ha[i]); v = fmin(fmax(v, -hb[j]), hb[j]);
    double pos[3];
    for (int k = 0; k < 3; k++) pos[k] = 0.5 * (ea[k] + u * A[i][k] + eb[k] + v * B[j][k]);
    set_contact(out, -beste, pos, en);
    return 1;
  }
  /* face contact: reference box owns the axis, incident box is clipped against it */
  int refA = code < 3, ax = refA ? code : code - 3;
  const double* pr = refA ? pa : pb; const double* pi_ = refA ? pb : pa;
  double(*Rr)[3] = refA ? A : B; double(*Ri)[3] = refA ? B : A;
  const double* hr = refA ? ha : hb; const double* hi = refA ? hb : ha;
  double nref[3];  /* outward normal of the reference face, pointing to the incident box */
  for (int k = 0; k < 3; k++) nref[k] = refA ? bestn[k] : -bestn[k];
  /* incident face: most anti-parallel to nref */
  int iax = 0; double mind = 1e300, isg = 1;
  for (int k = 0; k < 3; k++) {
    double dd = ho_dot3(Ri[k], nref);
    if (-fabs(dd) < mind) { mind = -fabs(dd); iax = k; isg = dd > 0 ? -1 : 1; }
  }
  int i1 = (iax + 1) % 3, i2 = (iax + 2) % 3, r1 = (ax + 1) % 3, r2 = (ax + 2) % 3;
  double fc[3];
  for (int k = 0; k < 3; k++) fc[k] = pi_[k] + isg * hi[iax] * Ri[iax][k] - pr[k];
  double poly[16][2], tmp[16][2];
  double sgs[4][2] = {{1, 1}, {-1, 1}, {-1, -1}, {1, -1}};
  double vz[4];
  for (int v = 0; v < 4; v++) {
    double w[3];
    for (int k = 0; k < 3; k++) w[k] = fc[k] + sgs[v][0] * hi[i1] * Ri[i1][k] + sgs[v][1] * hi[i2] * Ri[i2][k];
    poly[v][0] = ho_dot3(w, Rr[r1]); poly[v][1] = ho_dot3(w, Rr[r2]); vz[v] = ho_dot3(w, nref);
  }
  /* depth is affine over the incident face: z = z0 + gx x + gy y in reference 2-D coords */
  double M2[2][2] = {{poly[1][0] - poly[0][0], poly[1][1] - poly[0][1]}, {poly[3][0] - poly[0][0], poly[3][1] - poly[0][1]}};
  double det = M2[0][0] * M2[1][1] - M2[0][1] * M2[1][0];
  double gx = 0, gy = 0;
  if (fabs(det) > 1e-14) {
    double dz1 = vz[1] - vz[0], dz3 = vz[3] - vz[0];
    gx = (dz1 * M2[1][1] - dz3 * M2[0][1]) / det; gy = (dz3 * M2[0][0] - dz1 * M2[1][0]) / det;
  }
  double z0 = vz[0] - gx * poly[0][0] - gy * poly[0][1];
  int n = 4;
  n = clip_poly(poly, n, 0, hr[r1], 1, tmp); if (!n) return 0;
  n = clip_poly(tmp, n, 0, hr[r1], -1, poly); if (!n) return 0;
  n = clip_poly(poly, n, 1, hr[r2], 1, tmp); if (!n) return 0;
  n = clip_poly(tmp, n, 1, hr[r2], -1, poly); if (!n) return 0;
  double depth[16]; int keep[16], nk = 0;
  for (int v = 0; v < n; v++) {
    double z = z0 + gx * poly[v][0] + gy * poly[v][1];
    depth[v] = hr[ax] - z;
    if (depth[v] > 0) keep[nk++] = v;
  }
  if (!nk) return 0;
  int sel[4], ns = 0;
  if (nk <= 4 || maxout < 4) { for (int k = 0; k < nk && k < maxout && k < 4; k++) sel[ns++] = keep[k]; }
  else {
    int d0 = 0;
    for (int k = 1; k < nk; k++) if (depth[keep[k]] > depth[keep[d0]]) d0 = k;
    for (int k = 0; k < 4; k++) sel[ns++] = keep[(d0 + (k * nk) / 4) % nk];
  }
  for (int s = 0; s < ns; s++) {
    int v = sel[s];
    double z = hr[ax] - depth[v], pos[3];
    for (int k = 0; k < 3; k++)
      pos[k] = pr[k] + poly[v][0] * Rr[r1][k] + poly[v][1] * Rr[r2][k] + (z + 0.5 * depth[v]) * nref[k];
    set_contact(out + s, -depth[v], pos, bestn);
  }
  return ns;
}

/* ---------------------------------------------------------------- convex mesh vs primitive.
 * The collision shape of a mesh geom is its convex hull, stored as vertices AND face planes n.x <= d (mesh
 * frame, hoic_amd/mjcf.py).  Queries are plain loops over those tables (no GJK/EPA iteration, no degenerate
 * simplices; the same loops run one pair per lane on the GPU):
 *   signed distance of a point  ~  max_f (n_f.x - d_f)   (exact inside the hull and in front of a face; a lower
 *   bound next to edges/vertices, off by less than the facet size times the angle defect),
 *   capsule: minimise that convex piecewise-linear function along the axis, end spheres + minimiser (<= 2 points),
 *   box: hull vertices inside the box and box corners inside the hull (<= 4 deepest),
 *   plane: deepest hull vertices (<= 3).
 * MuJoCo uses libccd (MPR) with one point per pair here; this is a from-scratch substitute (PARITY UNPINNED). */
typedef struct { const double (*v)[3]; int nv; const double (*pl)[4]; int np; } hull_t;
static hull_t get_hull(const ho_model* m, int mesh) {
  hull_t h = {(const double(*)[3])m->mesh_vert[m->mesh_vertadr[mesh]], m->mesh_vertnum[mesh],
              (const double(*)[4])m->mesh_plane[m->mesh_planeadr[mesh]], m->mesh_planenum[mesh]};
  return h;
}
/* max over faces of alpha_f + t beta_f; returns the value and the arg-max face */
static double hull_line_max(const hull_t* h, const double a[3], const double d[3], double t, int* face) {
  double best = -1e300; int bf = 0;
  for (int f = 0; f < h->np; f++) {
    const double* p = h->pl[f];
    double v = p[0] * (a[0] + t * d[0]) + p[1] * (a[1] + t * d[1]) + p[2] * (a[2] + t * d[2]) - p[3];
    if (v > best) { best = v; bf = f; }
  }
  *face = bf;
  return best;
}
static int capsule_mesh(const ho_model* m, const double cp[3], const double cR[9], const double cs[3],
                        const double mp[3], const double mR[9], int mesh, ho_contact* out) {
  hull_t h = get_hull(m, mesh);
  double ax[3], rel[3], pc[3], al[3], a[3], d[3];
  col(cR, 2, ax);
  for (int i = 0; i < 3; i++) rel[i] = cp[i] - mp[i];
  mtv(mR, rel, pc); mtv(mR, ax, al);
  for (int i = 0; i < 3; i++) { a[i] = pc[i] - cs[1] * al[i]; d[i] = 2 * cs[1] * al[i]; }
  const double r = cs[0];
  /* minimise phi(t) = max_f (alpha_f + t beta_f) on [0,1]: bracket by the active faces at the ends, then
     intersect the two bracketing lines until the intersection is on the envelope */
  int f0, f1, fm;
  double v0 = hull_line_max(&h, a, d, 0, &f0), v1 = hull_line_max(&h, a, d, 1, &f1);
  double s0 = h.pl[f0][0] * d[0] + h.pl[f0][1] * d[1] + h.pl[f0][2] * d[2];
  double s1 = h.pl[f1][0] * d[0] + h.pl[f1][1] * d[1] + h.pl[f1][2] * d[2];
  double ts, vs; int fs;
  double nmin[3];     /* outward hull normal used at the minimiser */
  if (s0 >= 0) { ts = 0; vs = v0; fs = f0; for (int i = 0; i < 3; i++) nmin[i] = h.pl[f0][i]; }
  else if (s1 <= 0) { ts = 1; vs = v1; fs = f1; for (int i = 0; i < 3; i++) nmin[i] = h.pl[f1][i]; }
  else {
    double tl = 0, vl = v0, sl = s0, tr = 1, vr = v1, sr = s1; int fl = f0, fr = f1;
    ts = 0; vs = v0; fs = f0;
    for (int it = 0; it < 16; it++) {
      double t = ((vr - sr * tr) - (vl - sl * tl)) / (sl - sr);
      t = fmin(fmax(t, tl), tr);
      double v = hull_line_max(&h, a, d, t, &fm);
      double lineval = vl + sl * (t - tl);
      ts = t; vs = v; fs = fm;
      if (v <= lineval + 1e-12) break;
      double sm = h.pl[fm][0] * d[0] + h.pl[fm][1] * d[1] + h.pl[fm][2] * d[2];
      if (sm < 0) { tl = t; vl = v; sl = sm; fl = fm; } else { tr = t; vr = v; sr = sm; fr = fm; }
    }
    /* the minimiser sits where the two bracketing faces tie: use the combination of their normals that is
       perpendicular to the axis (zero sub-gradient), which is continuous in the pose */
    double lam = sr / (sr - sl);
    for (int i = 0; i < 3; i++) nmin[i] = lam * h.pl[fl][i] + (1 - lam) * h.pl[fr][i];
    ho_normalize3(nmin);
  }
  double tc[3], vc[3], nc3[3][3]; int nc = 0;
  if (v0 < r) { tc[nc] = 0; vc[nc] = v0; for (int i = 0; i < 3; i++) nc3[nc][i] = h.pl[f0][i]; nc++; }
  if (v1 < r) { tc[nc] = 1; vc[nc] = v1; for (int i = 0; i < 3; i++) nc3[nc][i] = h.pl[f1][i]; nc++; }
  if (nc < 2 && vs < r) {
    int dup = 0;
    for (int k = 0; k < nc; k++) if (fabs(ts - tc[k]) < 1e-6) dup = 1;
    if (!dup) { tc[nc] = ts; vc[nc] = vs; for (int i = 0; i < 3; i++) nc3[nc][i] = nmin[i]; nc++; }
  }
  (void)fs;
  int cnt = 0;
  for (int k = 0; k < nc && cnt < 2; k++) {
    const double* pl = nc3[k];
    double c[3], pos[3], nrm[3], pw[3], nw[3];
    for (int i = 0; i < 3; i++) { c[i] = a[i] + tc[k] * d[i]; nrm[i] = -pl[i]; pos[i] = c[i] - pl[i] * 0.5 * (r + vc[k]); }
    mv(mR, pos, pw); mv(mR, nrm, nw);
    for (int i = 0; i < 3; i++) pw[i] += mp[i];
    set_contact(out + cnt, vc[k] - r, pw, nw);
    cnt++;
  }
  return cnt;
}
static void keep_deepest(ho_contact* out, int* n, int cap, double dist, const double pos[3], const double nrm[3]) {
  int slot = -1;
  if (*n < cap) slot = (*n)++;
  else {
    int w = 0;
    for (int k = 1; k < cap; k++) if (out[k].dist > out[w].dist) w = k;
    if (dist < out[w].dist) slot = w;
  }
  if (slot >= 0) set_contact(out + slot, dist, pos, nrm);
}
static int box_mesh(const ho_model* m, const double bp[3], const double bR[9], const double bh[3],
                    const double mp[3], const double mR[9], int mesh, ho_contact* out) {
  hull_t h = get_hull(m, mesh);
  int n = 0;
  /* hull vertices inside the box */
  for (int v = 0; v < h.nv; v++) {
    double wv[3], rel[3], p[3];
    mv(mR, h.v[v], wv);
    for (int i = 0; i < 3; i++) { wv[i] += mp[i]; rel[i] = wv[i] - bp[i]; }
    mtv(bR, rel, p);
    double depth = 1e300; int k = -1;
    for (int i = 0; i < 3; i++) {
      double dd = bh[i] - fabs(p[i]);
      if (dd < depth) { depth = dd; k = i; }
    }
    if (depth <= 0) continue;
    double nl[3] = {0, 0, 0}, nw[3], pos[3];
    nl[k] = p[k] >= 0 ? 1 : -1;
    mv(bR, nl, nw);
    for (int i = 0; i < 3; i++) pos[i] = wv[i] + nw[i] * 0.5 * depth;
    keep_deepest(out, &n, 4, -depth, pos, nw);
  }
  /* box corners inside the hull */
  for (int c = 0; c < 8; c++) {
    double loc[3] = {(c & 1 ? bh[0] : -bh[0]), (c & 2 ? bh[1] : -bh[1]), (c & 4 ? bh[2] : -bh[2])}, wc[3], rel[3], p[3];
    mv(bR, loc, wc);
    for (int i = 0; i < 3; i++) { wc[i] += bp[i]; rel[i] = wc[i] - mp[i]; }
    mtv(mR, rel, p);
    double zero[3] = {0, 0, 0}; int f;
    double s = hull_line_max(&h, p, zero, 0, &f);
    if (s >= 0) continue;
    double nl[3] = {-h.pl[f][0], -h.pl[f][1], -h.pl[f][2]}, nw[3], pos[3];
    mv(mR, nl, nw);
    for (int i = 0; i < 3; i++) pos[i] = wc[i] - nw[i] * 0.5 * (-s);
    keep_deepest(out, &n, 4, s, pos, nw);
  }
  return n;
}

/* ---------------------------------------------------------------- dispatcher */
int ho_collide_pair(const ho_model* m, const ho_data* d, int pair, ho_contact* out, int maxout) {
  int g1 = m->pair_geom1[pair], g2 = m->pair_geom2[pair];
  int t1 = m->geom_type[g1], t2 = m->geom_type[g2];
  const double *p1 = d->geom_xpos[g1], *R1 = d->geom_xmat[g1], *s1 = m->geom_size[g1];
  const double *p2 = d->geom_xpos[g2], *R2 = d->geom_xmat[g2], *s2 = m->geom_size[g2];
  if (t1 == HOIC_GEOM_PLANE && t2 == HOIC_GEOM_CAPSULE) return plane_capsule(p1, R1, p2, R2, s2, out);
  if (t1 == HOIC_GEOM_PLANE && t2 == HOIC_GEOM_BOX) return plane_box(p1, R1, p2, R2, s2, out);
  if (t1 == HOIC_GEOM_PLANE && t2 == HOIC_GEOM_MESH) return plane_mesh(m, p1, R1, p2, R2, m->geom_meshid[g2], out);
  if (t1 == HOIC_GEOM_CAPSULE && t2 == HOIC_GEOM_CAPSULE) return capsule_capsule(p1, R1, s1, p2, R2, s2, out);
  if (t1 == HOIC_GEOM_CAPSULE && t2 == HOIC_GEOM_BOX) return capsule_box(p1, R1, s1, p2, R2, s2, out);
  if (t1 == HOIC_GEOM_BOX && t2 == HOIC_GEOM_BOX) return box_box(p1, R1, s1, p2, R2, s2, out, maxout);
  if (t1 == HOIC_GEOM_CAPSULE && t2 == HOIC_GEOM_MESH) return capsule_mesh(m, p1, R1, s1, p2, R2, m->geom_meshid[g2], out);
  if (t1 == HOIC_GEOM_BOX && t2 == HOIC_GEOM_MESH) return box_mesh(m, p1, R1, s1, p2, R2, m->geom_meshid[g2], out);
  return 0;
}
