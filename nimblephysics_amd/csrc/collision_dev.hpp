// collision_dev.hpp — per-lane box-box narrow phase (device).
//
// Same algorithm as the reference's DART detector for box pairs (dart/collision/dart/DARTCollide.cpp:
// 764-1450 dBoxBox: 15-axis SAT with the 1.05 fudge factor favouring face contacts, incident-face
// clipping against the reference face :512-575, up to 8 points kept, per-point type VERTEX_FACE /
// FACE_VERTEX / EDGE_EDGE with edge annotations :1280-1381, pure edge-edge case :1014-1054), written
// for one world per lane: every lane runs its own branch pattern, the clip polygons live in small
// private arrays, results are appended to the lane's contact list in HBM.
// Attribution: the algorithm restated here derives from the Open Dynamics Engine (ODE), Copyright (C) 2001-2003 Russell L. Smith, which the
// reference vendors under ODE's BSD-style licence (dart/external/odelcpsolver/, dart/collision/dart/DARTCollide.cpp); this file is an
// independent restatement for another execution model - ODE's arithmetic order and, where the bit-for-bit tests need them recognisable,
// its identifiers are kept on purpose.
#pragma once
#include "spatial_dev.hpp"

namespace NBL_NS {

constexpr int CT_VERTEX_FACE = 1, CT_FACE_VERTEX = 2, CT_EDGE_EDGE = 3, CT_SPHERE_BOX = 4, CT_BOX_SPHERE = 5, CT_SPHERE_SPHERE = 6;   // Contact.hpp:45-61
constexpr int CT_PIPE_SPHERE = 13, CT_SPHERE_PIPE = 14, CT_PIPE_PIPE = 15;   // capsule contacts, Contact.hpp:72-74

struct DevContact {
  V3 point, normal;
  double depth;
  int type;
  V3 edgeAClosest, edgeAFixed, edgeADir, edgeBClosest, edgeBFixed, edgeBDir;
};

DEV V3 colOf(const M3& R, int j) { return mk3(R.m[j], R.m[3 + j], R.m[6 + j]); }
DEV double norm3(V3 a) { return sqrt(dot(a, a)); }
DEV V3 unit3(V3 a) { double n = norm3(a); return mk3(a.x / n, a.y / n, a.z / n); }   // like Eigen's normalized(): divisions, no reciprocal
DEV void set3(V3& a, int i, double v) { if (i == 0) a.x = v; else if (i == 1) a.y = v; else a.z = v; }

// A small per-lane array in LDS, element i of the lane at base[i * ls] (ls = 64 or the narrow-phase lanes of the workgroup; conflict-free): the clip polygons are indexed
// dynamically, which in private memory means scratch (= global memory) round trips.
struct LaneBuf {
  double* base;
  int ls = 64;        // lanes that share the buffer (the stride between two elements of one lane)
  DEV double& operator[](int i) const { return base[i * ls]; }
  DEV LaneBuf at(int i) const { LaneBuf r; r.base = base + i * ls; r.ls = ls; return r; }
};

// clip the quad p against |x| <= h0, |y| <= h1; returns the number of points written to ret (<= 8).
// bufA, bufB, ret: 16 doubles each
DEV int intersectRectQuad(double h0, double h1, const double* p, LaneBuf bufA, LaneBuf bufB, LaneBuf ret) {
  for (int i = 0; i < 8; i++) bufA[i] = p[i];
  LaneBuf q = bufA, r = bufB;
  int nq = 4, nr = 0;
  bool done = false;
  for (int dir = 0; dir <= 1 && !done; dir++) {
    const double h = dir == 0 ? h0 : h1;
    for (int sign = -1; sign <= 1 && !done; sign += 2) {
      nr = 0;
      for (int i = 0; i < nq; i++) {
        const LaneBuf pq = q.at(2 * i);
        const LaneBuf nx = (i + 1 < nq) ? q.at(2 * i + 2) : q;
        const double pq0 = pq[0], pq1 = pq[1], nx0 = nx[0], nx1 = nx[1];
        const double pqd = dir == 0 ? pq0 : pq1, nxd = dir == 0 ? nx0 : nx1, pqo = dir == 0 ? pq1 : pq0, nxo = dir == 0 ? nx1 : nx0;
        bool in0 = sign * pqd < h, in1 = sign * nxd < h;
        if (in0) {
          r[2 * nr] = pq0; r[2 * nr + 1] = pq1;
          nr++;
          if (nr & 8) { done = true; break; }
        }
        if (in0 != in1) {
          r[2 * nr + (1 - dir)] = pqo + (nxo - pqo) / (nxd - pqd) * (sign * h - pqd);
          r[2 * nr + dir] = sign * h;
          nr++;
          if (nr & 8) { done = true; break; }
        }
      }
      LaneBuf t = q; q = r; r = t;
      nq = nr;
    }
  }
  for (int i = 0; i < nr * 2; i++) { const double v = q[i]; ret[i] = v; }
  return nr;
}

// Calls emit(contact) for every contact point (at most 8), in the reference's order; returns their number.  The contacts are
// handed over one at a time so that they stay in registers (an out[8] array of 26-double records lived in scratch memory).
template <class Emit>
DEV int boxBox(const T12& T1, V3 A, const T12& T2, V3 Bh, double clippingDepth, LaneBuf clip, Emit emit) {
  NBL_PHASE_FIRST(61);
  const double fudge = 1.05;
  const M3& R1 = T1.R;
  const M3& R2 = T2.R;
  V3 p = T2.p - T1.p;
  V3 pp = tmul(R1, p);
  M3 R = mulAtB(R1, R2), Q;
#pragma unroll
  for (int i = 0; i < 9; i++) Q.m[i] = fabs(R.m[i]);
  double Aa[3] = {A.x, A.y, A.z}, Bb[3] = {Bh.x, Bh.y, Bh.z}, ppa[3] = {pp.x, pp.y, pp.z};
  double s = -1e12;
  int code = 0, normalBox = 0, normalCol = 0;
  bool invert = false;
  V3 normalC = mk3(0, 0, 0);
  for (int k = 0; k < 3; k++) {
    double e1 = ppa[k], e2 = Aa[k] + Bb[0] * Q.m[3 * k] + Bb[1] * Q.m[3 * k + 1] + Bb[2] * Q.m[3 * k + 2];
    double s2 = fabs(e1) - e2;
    if (s2 > s) { s = s2; normalBox = 1; normalCol = k; invert = e1 < 0; code = k + 1; }
  }
  for (int k = 0; k < 3; k++) {
    double e1 = dot(colOf(R2, k), p), e2 = Aa[0] * Q.m[k] + Aa[1] * Q.m[3 + k] + Aa[2] * Q.m[6 + k] + Bb[k];
    double s2 = fabs(e1) - e2;
    if (s2 > s) { s = s2; normalBox = 2; normalCol = k; invert = e1 < 0; code = k + 4; }
  }
  for (int i = 0; i < 3; i++) {
    const int i1 = (i == 0) ? 1 : 0, i2 = (i == 2) ? 1 : 2;
    for (int j = 0; j < 3; j++) {
      const int j1 = (j == 0) ? 1 : 0, j2 = (j == 2) ? 1 : 2;
      V3 n;
      double e1;
      const double r0 = R.m[j], r1 = R.m[3 + j], r2 = R.m[6 + j];
      if (i == 0) { n = mk3(0, -r2, r1); e1 = ppa[2] * r1 - ppa[1] * r2; }
      else if (i == 1) { n = mk3(r2, 0, -r0); e1 = ppa[0] * r2 - ppa[2] * r0; }
      else { n = mk3(-r1, r0, 0); e1 = ppa[1] * r0 - ppa[0] * r1; }
      double e2 = Aa[i1] * Q.m[3 * i2 + j] + Aa[i2] * Q.m[3 * i1 + j] + Bb[j1] * Q.m[3 * i + j2] + Bb[j2] * Q.m[3 * i + j1];
      double s2 = fabs(e1) - e2;
      double l = sqrt(n.x * n.x + n.y * n.y + n.z * n.z);
      if (l > 0) {
        s2 /= l;
        if (s2 * fudge > s) { s = s2; normalBox = 0; normalC = mk3(n.x / l, n.y / l, n.z / l); invert = e1 < 0; code = 7 + 3 * i + j; }
      }
    }
  }
  NBL_PHASE_FIRST(62);
  if (!code) return 0;
  if (s > 0.0) return 0;
  V3 normal;
  if (normalBox == 1) normal = colOf(R1, normalCol);
  else if (normalBox == 2) normal = colOf(R2, normalCol);
  else normal = unit3(mul(R1, normalC));
  if (invert) normal = -normal;

  if (code > 6) {
    V3 pa = T1.p;
    for (int j = 0; j < 3; j++) {
      double sign = (dot(normal, colOf(R1, j)) > -1e-10) ? 1.0 : -1.0;
      pa = pa + (sign * Aa[j]) * colOf(R1, j);
    }
    V3 pb = T2.p;
    for (int j = 0; j < 3; j++) {
      double sign = (dot(normal, colOf(R2, j)) > -1e-3) ? -1.0 : 1.0;
      pb = pb + (sign * Bb[j]) * colOf(R2, j);
    }
    V3 ua = colOf(R1, (code - 7) / 3), ub = colOf(R2, (code - 7) % 3);
    V3 dp = pb - pa;
    double uaub = dot(ua, ub), q1 = dot(ua, dp), q2 = -dot(ub, dp), d = 1 - uaub * uaub, alpha = 0, beta = 0;
    if (d > 0) { d = 1.0 / d; alpha = (q1 + uaub * q2) * d; beta = (uaub * q1 + q2) * d; }
    V3 fixedA = pa, fixedB = pb;
    pa = pa + alpha * ua;
    pb = pb + beta * ub;
    double pen = -s;
    if (pen > clippingDepth) return 0;
    DevContact c;
    c.point = 0.5 * (pa + pb);
    c.normal = -normal;
    c.depth = pen;
    c.type = CT_EDGE_EDGE;
    c.edgeAClosest = pa; c.edgeAFixed = fixedA; c.edgeADir = unit3(ua);
    c.edgeBClosest = pb; c.edgeBFixed = fixedB; c.edgeBDir = unit3(ub);
    emit(c);
    return 1;
  }

  const bool flip = code > 3;
  const M3& Ra = flip ? R2 : R1;
  const M3& Rb = flip ? R1 : R2;
  V3 pa = flip ? T2.p : T1.p, pb = flip ? T1.p : T2.p;
  const double* Sa = flip ? Bb : Aa;
  const double* Sb = flip ? Aa : Bb;
  V3 normal2 = flip ? -normal : normal;
  V3 nr = tmul(Rb, normal2);
  double anr0 = fabs(nr.x), anr1 = fabs(nr.y), anr2 = fabs(nr.z);
  int lanr, a1, a2;
  if (anr1 > anr0) {
    if (anr1 > anr2) { a1 = 0; lanr = 1; a2 = 2; } else { a1 = 0; a2 = 1; lanr = 2; }
  } else {
    if (anr0 > anr2) { lanr = 0; a1 = 1; a2 = 2; } else { a1 = 0; a2 = 1; lanr = 2; }
  }
  V3 center;
  if (get(nr, lanr) < 0) center = pb - pa + Sb[lanr] * colOf(Rb, lanr);
  else center = pb - pa - Sb[lanr] * colOf(Rb, lanr);
  const int codeN = flip ? code - 4 : code - 1;
  int code1, code2;
  if (codeN == 0) { code1 = 1; code2 = 2; } else if (codeN == 1) { code1 = 0; code2 = 2; } else { code1 = 0; code2 = 1; }
  double quad[8];
  double c1 = dot(center, colOf(Ra, code1)), c2 = dot(center, colOf(Ra, code2));
  double m11 = dot(colOf(Ra, code1), colOf(Rb, a1)), m12 = dot(colOf(Ra, code1), colOf(Rb, a2));
  double m21 = dot(colOf(Ra, code2), colOf(Rb, a1)), m22 = dot(colOf(Ra, code2), colOf(Rb, a2));
  {
    double k1 = m11 * Sb[a1], k2 = m21 * Sb[a1], k3 = m12 * Sb[a2], k4 = m22 * Sb[a2];
    quad[0] = c1 - k1 - k3; quad[1] = c2 - k2 - k4;
    quad[2] = c1 - k1 + k3; quad[3] = c2 - k2 + k4;
    quad[4] = c1 + k1 + k3; quad[5] = c2 + k2 + k4;
    quad[6] = c1 + k1 - k3; quad[7] = c2 + k2 - k4;
  }
  const double rect0 = Sa[code1], rect1 = Sa[code2];
  const LaneBuf ret = clip.at(32);   // clip: 48 doubles per lane (two polygon buffers + the result)
  NBL_PHASE_FIRST(39);
  int n = intersectRectQuad(rect0, rect1, quad, clip, clip.at(16), ret);
  NBL_PHASE_FIRST(63);
  if (n < 1) return 0;
  double det1 = 1.0 / (m11 * m22 - m12 * m21);
  m11 *= det1; m12 *= det1; m21 *= det1; m22 *= det1;
  V3 otherNormal = colOf(Rb, lanr);
  if (dot(otherNormal, normal) < 0) otherNormal = -otherNormal;
  V3 ortho1 = colOf(Rb, a1), ortho2 = colOf(Rb, a2);
  V3 faceCenter = pb - Sb[lanr] * otherNormal;
  int cnum = 0;
  for (int j = 0; j < n; j++) {
    double k1 = m22 * (ret[j * 2] - c1) - m12 * (ret[j * 2 + 1] - c2);
    double k2 = -m21 * (ret[j * 2] - c1) + m11 * (ret[j * 2 + 1] - c2);
    V3 pt = center + k1 * colOf(Rb, a1) + k2 * colOf(Rb, a2);
    double dep = Sa[codeN] - dot(normal2, pt);
    if (!(dep >= 0)) continue;
    const double x = ret[j * 2], y = ret[j * 2 + 1];
    DevContact c;
    c.point = pt + pa;
    c.normal = -normal;
    c.depth = dep;
    c.edgeAClosest = c.edgeAFixed = c.edgeADir = c.edgeBClosest = c.edgeBFixed = c.edgeBDir = mk3(0, 0, 0);
    const bool onX = fabs(x) == rect0, onY = fabs(y) == rect1;
    if (onX && onY) {
      if (flip) { c.type = CT_FACE_VERTEX; c.point = c.point + c.depth * c.normal; }
      else { c.type = CT_VERTEX_FACE; c.point = c.point - c.depth * c.normal; }
    } else if (!onX && !onY) {
      c.type = flip ? CT_VERTEX_FACE : CT_FACE_VERTEX;
    } else {
      c.type = CT_EDGE_EDGE;
      double faceX = x > 0 ? rect0 : -rect0, faceY = y > 0 ? rect1 : -rect1;
      V3 faceCenterA = pa + Sa[codeN] * normal;
      c.edgeAFixed = faceCenterA + faceX * colOf(Ra, code1) + faceY * colOf(Ra, code2);
      c.edgeADir = unit3(c.point - c.edgeAFixed);
      c.edgeAClosest = c.point;
      double incX = dot(ortho1, c.point) - dot(ortho1, pb), incY = dot(ortho2, c.point) - dot(ortho2, pb);
      double signX = incX == 0 ? 1.0 : (incX / fabs(incX)), signY = incY == 0 ? 1.0 : (incY / fabs(incY));
      V3 nearestB = (signX * Sb[a1]) * ortho1 + (signY * Sb[a2]) * ortho2 + faceCenter;
      double distX = fabs(fabs(incX) - Sb[a1]), distY = fabs(fabs(incY) - Sb[a2]);
      V3 otherB;
      if (distX < distY) otherB = (signX * Sb[a1]) * ortho1 + (-1 * signY * Sb[a2]) * ortho2 + faceCenter;
      else otherB = (-1 * signX * Sb[a1]) * ortho1 + (signY * Sb[a2]) * ortho2 + faceCenter;
      c.edgeBDir = unit3(nearestB - otherB);
      c.edgeBFixed = nearestB;
      if (flip) {
        V3 t = c.edgeADir; c.edgeADir = c.edgeBDir; c.edgeBDir = t;
        t = c.edgeAFixed; c.edgeAFixed = c.edgeBFixed; c.edgeBFixed = t;
      }
    }
    emit(c);
    cnum++;
  }
  return cnum;
}

// collideBoxSphere (DARTCollide.cpp:1482-1653; the box is the first object) / collideSphereBox (:1655-1810; the sphere is):
// the sphere centre is clamped to the box, every clamped axis locks that face normal; the normal points from the second
// object towards the first; a centre inside the box is a plain FACE_VERTEX / VERTEX_FACE contact at the centre.
// Annotations in the edge slots: edgeAFixed = sphere centre, edgeADir / edgeBFixed / edgeBDir = the locked face normals.
template <class Emit>
DEV int sphereBoxPair(bool sphereFirst, double r, const T12& Ts, V3 half, const T12& Tb, double clippingDepth, Emit emit) {
  const V3 c0 = Ts.p;
  const V3 pl = tmul(Tb.R, c0 - Tb.p);
  const double px = fmin(fmax(pl.x, -half.x), half.x), py = fmin(fmax(pl.y, -half.y), half.y), pz = fmin(fmax(pl.z, -half.z), half.z);
  const bool lx = pl.x < -half.x || pl.x > half.x, ly = pl.y < -half.y || pl.y > half.y, lz = pl.z < -half.z || pl.z > half.z;
  const bool inside = !(lx || ly || lz);
  DevContact c;
  c.edgeAClosest = c.edgeBClosest = mk3(0, 0, 0);
  c.edgeAFixed = c0;
  c.edgeADir = lx ? colOf(Tb.R, 0) : mk3(0, 0, 0);
  c.edgeBFixed = ly ? colOf(Tb.R, 1) : mk3(0, 0, 0);
  c.edgeBDir = lz ? colOf(Tb.R, 2) : mk3(0, 0, 0);
  c.type = sphereFirst ? CT_SPHERE_BOX : CT_BOX_SPHERE;
  const double outward = sphereFirst ? 1.0 : -1.0;
  // nearest side of the (clamped) point, in the reference's comparison order
  double mn = half.x - fabs(px);
  int idx = 0;
  double t = half.y - fabs(py);
  if (t < mn) { mn = t; idx = 1; }
  t = half.z - fabs(pz);
  if (t < mn) { mn = t; idx = 2; }
  const double pidx = idx == 0 ? px : (idx == 1 ? py : pz);
  const V3 sideNormal = ((pidx > 0.0 ? 1.0 : -1.0) * outward) * colOf(Tb.R, idx);
  if (inside) {
    const double pen = mn + r;
    if (pen > clippingDepth) return 0;
    c.type = sphereFirst ? CT_VERTEX_FACE : CT_FACE_VERTEX;
    c.point = c0; c.normal = sideNormal; c.depth = pen;
    emit(c);
    return 1;
  }
  const V3 contactpt = mul(Tb.R, mk3(px, py, pz)) + Tb.p;
  V3 normal = sphereFirst ? c0 - contactpt : contactpt - c0;
  const double mag = norm3(normal), pen = r - mag;
  if (pen > clippingDepth) return 0;
  if (pen < 0.0) return 0;
  if (mag > 1e-6) normal = (1.0 / mag) * normal;     // DART_COLLISION_EPS
  else normal = sideNormal;
  c.point = contactpt; c.normal = normal; c.depth = pen;
  emit(c);
  return 1;
}

// collideSphereSphere (DARTCollide.cpp:1812-1880).  Annotations: edgeAFixed = centre A, edgeBFixed = centre B, edgeADir = (rA, rB, 0)
template <class Emit>
DEV int sphereSphere(double r0, const T12& T0, double r1, const T12& T1, double clippingDepth, Emit emit) {
  const double rsum = r0 + r1;
  V3 normal = T0.p - T1.p;
  const double nsq = dot(normal, normal);
  if (nsq > rsum * rsum) return 0;
  DevContact c;
  c.edgeAClosest = c.edgeBClosest = c.edgeBDir = mk3(0, 0, 0);
  c.type = CT_SPHERE_SPHERE;
  c.edgeAFixed = T0.p; c.edgeBFixed = T1.p;
  const double w0 = r0 / rsum, w1 = r1 / rsum;
  c.edgeADir = mk3(w0 * rsum, w1 * rsum, 0);
  c.point = w1 * T0.p + w0 * T1.p;
  if (nsq < 1e-6) { c.normal = mk3(0, 0, 0); c.depth = rsum; }
  else { const double len = sqrt(nsq); c.normal = (1.0 / len) * normal; c.depth = rsum - len; }
  if (c.depth > clippingDepth) return 0;
  emit(c);
  return 1;
}

// ---- capsules (CapsuleShape: axis = z of the shape frame, `height` = length of the cylinder part) --------------------------------
// dSegmentsClosestApproach (DARTCollide.cpp:301-381): parameters of the closest points of the segments pa->pb and ua->ub
DEV void segmentsClosestApproach(V3 pa, V3 ua, V3 pb, V3 ub, double* alpha, double* beta) {
  const V3 u = pb - pa, v = ub - ua, w = pa - ua;
  const double a = dot(u, u), b = dot(u, v), c = dot(v, v), d = dot(u, w), e = dot(v, w);
  const double D = a * c - b * b;
  double sN, sD = D, tN, tD = D;
  const double SMALL_NUM = 1e-15;
  if (D < SMALL_NUM) { sN = 0.0; sD = 1.0; tN = e; tD = c; }
  else {
    sN = (b * e - c * d);
    tN = (a * e - b * d);
    if (sN < 0.0) { sN = 0.0; tN = e; tD = c; }
    else if (sN > sD) { sN = sD; tN = e + b; tD = c; }
  }
  if (tN < 0.0) {
    tN = 0.0;
    if (-d < 0.0) sN = 0.0;
    else if (-d > a) sN = sD;
    else { sN = -d; sD = a; }
  } else if (tN > tD) {
    tN = tD;
    if ((-d + b) < 0.0) sN = 0;
    else if ((-d + b) > a) sN = sD;
    else { sN = (-d + b); sD = a; }
  }
  *alpha = (fabs(sN) < SMALL_NUM ? 0.0 : sN / sD);
  *beta = (fabs(tN) < SMALL_NUM ? 0.0 : tN / tD);
}

// dDistPointToSegment (DARTCollide.cpp:384-410)
DEV double distPointToSegment(V3 p, V3 ua, V3 ub, double* alpha) {
  const V3 v = ub - ua, w = p - ua;
  const double c1 = dot(w, v);
  if (c1 <= 0) { *alpha = 0; return norm3(p - ua); }
  const double c2 = dot(v, v);
  if (c2 <= c1) { *alpha = 1; return norm3(p - ub); }
  *alpha = c1 / c2;
  return norm3(p - (ua + *alpha * v));
}

// Annotations of the capsule contact types in the four edge slots (the backward pass reads the two capsule radii from the model):
//   SPHERE_SPHERE (an end cap against an end cap / a sphere): like sphereSphere above
//   SPHERE_PIPE / PIPE_SPHERE: edgeAFixed = sphere centre, edgeADir = pipe direction, edgeBFixed = the pipe's fixed point,
//                              edgeBDir = (|closest point on the pipe - sphere centre|, sphere radius, pipe radius)
//   PIPE_PIPE: edgeAFixed / edgeADir / edgeBFixed / edgeBDir = the two axis lines, as for EDGE_EDGE
// collideCapsuleCapsule (DARTCollide.cpp:4183-4284)
template <class Emit>
DEV int capsuleCapsule(double height0, double radius0, const T12& T0, double height1, double radius1, const T12& T1,
                       double clippingDepth, Emit emit) {
  const V3 z0 = colOf(T0.R, 2), z1 = colOf(T1.R, 2);
  const V3 pa = (-(height0 / 2)) * z0 + T0.p, pb = (height0 / 2) * z0 + T0.p;
  const V3 ua = (-(height1 / 2)) * z1 + T1.p, ub = (height1 / 2) * z1 + T1.p;
  double alpha, beta;
  segmentsClosestApproach(pa, ua, pb, ub, &alpha, &beta);
  alpha = fmin(fmax(alpha, 0.0), 1.0);
  beta = fmin(fmax(beta, 0.0), 1.0);
  const V3 closest0 = pa + alpha * (pb - pa), closest1 = ua + beta * (ub - ua);
  const double dist = norm3(closest0 - closest1), rsum = radius0 + radius1;
  if (!(dist <= rsum)) return 0;
  radius0 /= rsum; radius1 /= rsum;
  DevContact c;
  c.edgeAClosest = closest0; c.edgeBClosest = closest1;
  c.depth = rsum - dist;
  if (c.depth > clippingDepth) return 0;
  c.point = radius1 * closest0 + radius0 * closest1;
  c.normal = unit3(closest0 - closest1);
  const double SPHERE_THRESHOLD = 1e-8;
  const bool isSphere0 = fabs(alpha) < SPHERE_THRESHOLD || fabs(1 - alpha) < SPHERE_THRESHOLD;
  const bool isSphere1 = fabs(beta) < SPHERE_THRESHOLD || fabs(1 - beta) < SPHERE_THRESHOLD;
  if (isSphere0 && isSphere1) {
    c.type = CT_SPHERE_SPHERE;
    c.edgeAFixed = closest0; c.edgeBFixed = closest1; c.edgeADir = mk3(radius0 * rsum, radius1 * rsum, 0); c.edgeBDir = mk3(0, 0, 0);
  } else if (isSphere0) {
    c.type = CT_SPHERE_PIPE;
    c.edgeAFixed = closest0; c.edgeADir = unit3(ub - ua); c.edgeBFixed = ua; c.edgeBDir = mk3(dist, radius0 * rsum, radius1 * rsum);
  } else if (isSphere1) {
    c.type = CT_PIPE_SPHERE;
    c.edgeAFixed = closest1; c.edgeADir = unit3(pb - pa); c.edgeBFixed = pa; c.edgeBDir = mk3(dist, radius1 * rsum, radius0 * rsum);
  } else {
    c.type = CT_PIPE_PIPE;
    c.edgeAFixed = pa; c.edgeADir = unit3(pb - pa); c.edgeBFixed = ua; c.edgeBDir = unit3(ub - ua);
  }
  emit(c);
  return 1;
}

// collideSphereCapsule (DARTCollide.cpp:4286-4352; the sphere is the first object) / collideCapsuleSphere (:4354-4420)
template <class Emit>
DEV int sphereCapsulePair(bool sphereFirst, double rSphere, const T12& Ts, double height, double rCapsule, const T12& Tc,
                          double clippingDepth, Emit emit) {
  double alpha;
  const V3 center = Ts.p, zc = colOf(Tc.R, 2);
  const V3 ua = (-(height / 2)) * zc + Tc.p, ub = (height / 2) * zc + Tc.p;
  const double dist = distPointToSegment(center, ua, ub, &alpha);
  double radius0 = sphereFirst ? rSphere : rCapsule, radius1 = sphereFirst ? rCapsule : rSphere;
  if (!(dist < radius0 + radius1)) return 0;
  const V3 closest = ua + alpha * (ub - ua);
  const double rsum = radius0 + radius1;
  radius0 /= rsum; radius1 /= rsum;
  DevContact c;
  c.edgeAClosest = c.edgeBClosest = mk3(0, 0, 0);
  c.depth = rsum - dist;
  if (c.depth > clippingDepth) return 0;
  const V3 first = sphereFirst ? center : closest, second = sphereFirst ? closest : center;
  c.point = radius1 * first + radius0 * second;
  c.normal = unit3(first - second);
  const double SPHERE_THRESHOLD = 1e-8;
  if (fabs(alpha) < SPHERE_THRESHOLD || fabs(1 - alpha) < SPHERE_THRESHOLD) {
    c.type = CT_SPHERE_SPHERE;
    c.edgeAFixed = first; c.edgeBFixed = second; c.edgeADir = mk3(radius0 * rsum, radius1 * rsum, 0); c.edgeBDir = mk3(0, 0, 0);
  } else {
    c.type = sphereFirst ? CT_SPHERE_PIPE : CT_PIPE_SPHERE;
    c.edgeAFixed = center; c.edgeADir = unit3(ub - ua); c.edgeBFixed = ua;
    c.edgeBDir = mk3(dist, (sphereFirst ? radius0 : radius1) * rsum, (sphereFirst ? radius1 : radius0) * rsum);
  }
  emit(c);
  return 1;
}

}  // namespace NBL_NS
