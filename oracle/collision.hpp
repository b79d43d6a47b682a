// oracle/collision.hpp — TEST INFRASTRUCTURE ONLY.
//
// Box-box narrow phase restated from the reference's DART collision detector
// (dart/collision/dart/DARTCollide.cpp:764-1450 `dBoxBox`, :422-600 `intersectRectQuad`,
// :271-300 `dLineClosestApproach`) plus the pair loop / duplicate filter of
// dart/collision/dart/DARTCollisionDetector.cpp:150-175, 360-400.
// Same separating-axis order, same fudge factor and tie-breaks, same clipping order, so contact
// order, count, types and edge annotations match the reference.
// Attribution: the algorithm restated here derives from the Open Dynamics Engine (ODE), Copyright (C) 2001-2003 Russell L. Smith, which the
// reference vendors under ODE's BSD-style licence (dart/external/odelcpsolver/, dart/collision/dart/DARTCollide.cpp); this file is an
// independent restatement for another execution model - ODE's arithmetic order and, where the bit-for-bit tests need them recognisable,
// its identifiers are kept on purpose.
#pragma once
#include <vector>

#include "dynamics.hpp"

namespace nbo {

enum ContactType { CT_UNSUPPORTED = 0, CT_VERTEX_FACE = 1, CT_FACE_VERTEX = 2, CT_EDGE_EDGE = 3,   // Contact.hpp:45-61
                   CT_SPHERE_BOX = 4, CT_BOX_SPHERE = 5, CT_SPHERE_SPHERE = 6,
                   CT_PIPE_SPHERE = 13, CT_SPHERE_PIPE = 14, CT_PIPE_PIPE = 15 };   // capsule contacts, Contact.hpp:72-74

struct Contact {
  Vec3 point, normal;
  s_t depth;
  int type;
  int bodyA, bodyB, boxA, boxB;
  Vec3 edgeAClosestPoint, edgeAFixedPoint, edgeADir, edgeBClosestPoint, edgeBFixedPoint, edgeBDir;
  // sphere contacts (Contact.hpp: sphereCenter, face{1,2,3}{Locked,Normal}, center{A,B}, radius{A,B})
  Vec3 sphereCenter = mk3(0, 0, 0), faceNormal[3] = {mk3(0, 0, 0), mk3(0, 0, 0), mk3(0, 0, 0)};
  bool faceLocked[3] = {false, false, false};
  Vec3 centerA = mk3(0, 0, 0), centerB = mk3(0, 0, 0);
  s_t radiusA = 0, radiusB = 0;
  // capsule contacts (Contact.hpp:186-199): the cylinder ("pipe") side is a line through pipeFixedPoint along pipeDir
  Vec3 pipeDir = mk3(0, 0, 0), pipeFixedPoint = mk3(0, 0, 0), pipeClosestPoint = mk3(0, 0, 0);
  s_t sphereRadius = 0, pipeRadius = 0;
};

inline Vec3 col(const Mat3& R, int j) { return mk3(R(0, j), R(1, j), R(2, j)); }
// Eigen's normalized() / normalize() (3.3 and later): every component DIVIDED by the norm, no reciprocal
inline Vec3 normalized(const Vec3& a) { s_t n = norm(a); return mk3(a[0] / n, a[1] / n, a[2] / n); }

// DARTCollide.cpp:271-300
inline void lineClosestApproach(const Vec3& pa, const Vec3& ua, const Vec3& pb, const Vec3& ub, s_t* alpha, s_t* beta) {
  Vec3 p = pb - pa;
  s_t uaub = dot(ua, ub), q1 = dot(ua, p), q2 = -dot(ub, p);
  s_t d = 1 - uaub * uaub;
  if (d <= 0) { *alpha = 0; *beta = 0; }
  else { d = 1.0 / d; *alpha = (q1 + uaub * q2) * d; *beta = (uaub * q1 + q2) * d; }
}

// DARTCollide.cpp:512-575: clip the quad p (4 points) against the rectangle |x|<=h[0], |y|<=h[1]
inline int intersectRectQuad(const s_t h[2], const s_t p[8], s_t ret[16]) {
  int nq = 4, nr = 0;
  s_t bufA[16], bufB[16];
  for (int i = 0; i < 8; i++) bufA[i] = p[i];
  s_t* q = bufA;
  s_t* r = bufB;
  bool done = false;
  for (int dir = 0; dir <= 1 && !done; dir++) {
    for (int sign = -1; sign <= 1 && !done; sign += 2) {
      nr = 0;
      for (int i = 0; i < nq && !done; i++) {
        const s_t* pq = q + 2 * i;
        const s_t* nextq = (i + 1 < nq) ? pq + 2 : q;
        bool in0 = sign * pq[dir] < h[dir], in1 = sign * nextq[dir] < h[dir];
        if (in0) {
          r[2 * nr] = pq[0]; r[2 * nr + 1] = pq[1];
          nr++;
          if (nr & 8) { done = true; break; }
        }
        if (in0 ^ in1) {
          r[2 * nr + (1 - dir)] = pq[1 - dir] + (nextq[1 - dir] - pq[1 - dir]) / (nextq[dir] - pq[dir]) * (sign * h[dir] - pq[dir]);
          r[2 * nr + dir] = sign * h[dir];
          nr++;
          if (nr & 8) { done = true; break; }
        }
      }
      // swap
      s_t* t = q; q = r; r = t;
      nq = nr;
    }
  }
  for (int i = 0; i < nr * 2; i++) ret[i] = q[i];
  return nr;
}

// dBoxBox.  T1/T2 world transforms of the box shapes, A/B half sizes.  Appends to `out`.
inline int boxBox(const Iso& T1, const Vec3& A, const Iso& T2, const Vec3& B, s_t clippingDepth, std::vector<Contact>& out) {
  const s_t fudge = 1.05;
  const Mat3& R1 = T1.R;
  const Mat3& R2 = T2.R;
  Vec3 p = T2.p - T1.p;
  Vec3 pp = tmul(R1, p);
  Mat3 R = transpose(R1) * R2, Q;
  for (int i = 0; i < 9; i++) Q.m[i] = std::fabs(R.m[i]);

  s_t s = -1e12;
  int code = 0, normalBox = 0, normalCol = 0;
  bool invert = false;
  Vec3 normalC = mk3(0, 0, 0);
  // face axes of box 1 (codes 1-3) and box 2 (codes 4-6): strict '>' keeps the first maximum
  for (int k = 0; k < 3; k++) {
    s_t e1 = pp[k], e2 = A[k] + B[0] * Q(k, 0) + B[1] * Q(k, 1) + B[2] * Q(k, 2);
    s_t s2 = std::fabs(e1) - e2;
    if (s2 > s) { s = s2; normalBox = 1; normalCol = k; invert = e1 < 0; code = k + 1; }
  }
  for (int k = 0; k < 3; k++) {
    s_t e1 = dot(col(R2, k), p), e2 = A[0] * Q(0, k) + A[1] * Q(1, k) + A[2] * Q(2, k) + B[k];
    s_t s2 = std::fabs(e1) - e2;
    if (s2 > s) { s = s2; normalBox = 2; normalCol = k; invert = e1 < 0; code = k + 4; }
  }
  // edge x edge axes u_i x v_j (codes 7..15), scaled, with the fudge factor favouring face contacts
  for (int i = 0; i < 3; i++) {
    int i1 = (i == 0) ? 1 : 0, i2 = (i == 2) ? 1 : 2;
    for (int j = 0; j < 3; j++) {
      int j1 = (j == 0) ? 1 : 0, j2 = (j == 2) ? 1 : 2;
      Vec3 n;
      s_t e1;
      if (i == 0) { n = mk3(0, -R(2, j), R(1, j)); e1 = pp[2] * R(1, j) - pp[1] * R(2, j); }
      else if (i == 1) { n = mk3(R(2, j), 0, -R(0, j)); e1 = pp[0] * R(2, j) - pp[2] * R(0, j); }
      else { n = mk3(-R(1, j), R(0, j), 0); e1 = pp[1] * R(0, j) - pp[0] * R(1, j); }
      s_t e2 = A[i1] * Q(i2, j) + A[i2] * Q(i1, j) + B[j1] * Q(i, j2) + B[j2] * Q(i, j1);
      s_t s2 = std::fabs(e1) - e2;
      s_t l = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
      if (l > 0) {
        s2 /= l;
        if (s2 * fudge > s) {
          s = s2; normalBox = 0; normalC = mk3(n[0] / l, n[1] / l, n[2] / l);   /* (n1) / l ..., DARTCollide.cpp:867-869 */ invert = e1 < 0; code = 7 + 3 * i + j;
        }
      }
    }
  }
  if (!code) return 0;
  if (s > 0.0) return 0;

  Vec3 normal;
  if (normalBox == 1) normal = col(R1, normalCol);
  else if (normalBox == 2) normal = col(R2, normalCol);
  else normal = normalized(R1 * normalC);
  if (invert) normal = -normal;

  if (code > 6) {
    // edge-edge: one contact at the midpoint of the closest points of the two edges
    Vec3 pa = T1.p;
    for (int j = 0; j < 3; j++) {
      s_t sign = (dot(normal, col(R1, j)) > -1e-10) ? 1.0 : -1.0;
      pa = pa + (sign * A[j]) * col(R1, j);
    }
    Vec3 pb = T2.p;
    for (int j = 0; j < 3; j++) {
      s_t sign = (dot(normal, col(R2, j)) > -1e-3) ? -1.0 : 1.0;
      pb = pb + (sign * B[j]) * col(R2, j);
    }
    Vec3 ua = col(R1, (code - 7) / 3), ub = col(R2, (code - 7) % 3);
    s_t alpha, beta;
    lineClosestApproach(pa, ua, pb, ub, &alpha, &beta);
    Vec3 fixedA = pa, fixedB = pb;
    pa = pa + alpha * ua;
    pb = pb + beta * ub;
    s_t penetration = -s;
    if (penetration > clippingDepth) return 0;
    Contact c;
    c.point = 0.5 * (pa + pb);
    c.normal = -normal;
    c.depth = penetration;
    c.type = CT_EDGE_EDGE;
    c.edgeAClosestPoint = pa; c.edgeAFixedPoint = fixedA; c.edgeADir = normalized(ua);
    c.edgeBClosestPoint = pb; c.edgeBFixedPoint = fixedB; c.edgeBDir = normalized(ub);
    out.push_back(c);
    return 1;
  }

  // face-something: reference face on box a, incident face on box b
  const Mat3 *Ra, *Rb;
  Vec3 pa, pb, Sa, Sb;
  bool flip;
  if (code <= 3) { Ra = &R1; Rb = &R2; pa = T1.p; pb = T2.p; Sa = A; Sb = B; flip = false; }
  else { Ra = &R2; Rb = &R1; pa = T2.p; pb = T1.p; Sa = B; Sb = A; flip = true; }
  Vec3 normal2 = (code <= 3) ? normal : -normal;
  Vec3 nr = tmul(*Rb, normal2);
  Vec3 anr = mk3(std::fabs(nr[0]), std::fabs(nr[1]), std::fabs(nr[2]));
  int lanr, a1, a2;
  if (anr[1] > anr[0]) {
    if (anr[1] > anr[2]) { a1 = 0; lanr = 1; a2 = 2; } else { a1 = 0; a2 = 1; lanr = 2; }
  } else {
    if (anr[0] > anr[2]) { lanr = 0; a1 = 1; a2 = 2; } else { a1 = 0; a2 = 1; lanr = 2; }
  }
  Vec3 center;
  if (nr[lanr] < 0) center = pb - pa + Sb[lanr] * col(*Rb, lanr);
  else center = pb - pa - Sb[lanr] * col(*Rb, lanr);
  int codeN = (code <= 3) ? code - 1 : code - 4, code1, code2;
  if (codeN == 0) { code1 = 1; code2 = 2; } else if (codeN == 1) { code1 = 0; code2 = 2; } else { code1 = 0; code2 = 1; }
  s_t quad[8];
  s_t c1 = dot(center, col(*Ra, code1)), c2 = dot(center, col(*Ra, code2));
  s_t m11 = dot(col(*Ra, code1), col(*Rb, a1)), m12 = dot(col(*Ra, code1), col(*Rb, a2));
  s_t m21 = dot(col(*Ra, code2), col(*Rb, a1)), m22 = dot(col(*Ra, code2), col(*Rb, a2));
  {
    s_t k1 = m11 * Sb[a1], k2 = m21 * Sb[a1], k3 = m12 * Sb[a2], k4 = m22 * Sb[a2];
    quad[0] = c1 - k1 - k3; quad[1] = c2 - k2 - k4;
    quad[2] = c1 - k1 + k3; quad[3] = c2 - k2 + k4;
    quad[4] = c1 + k1 + k3; quad[5] = c2 + k2 + k4;
    quad[6] = c1 + k1 - k3; quad[7] = c2 + k2 - k4;
  }
  s_t rect[2] = {Sa[code1], Sa[code2]};
  s_t ret[16];
  int n = intersectRectQuad(rect, quad, ret);
  if (n < 1) return 0;
  s_t point[24], dep[8];
  s_t det1 = 1.0 / (m11 * m22 - m12 * m21);
  m11 *= det1; m12 *= det1; m21 *= det1; m22 *= det1;
  int cnum = 0;
  for (int j = 0; j < n; j++) {
    s_t k1 = m22 * (ret[j * 2] - c1) - m12 * (ret[j * 2 + 1] - c2);
    s_t k2 = -m21 * (ret[j * 2] - c1) + m11 * (ret[j * 2 + 1] - c2);
    Vec3 pt = center + k1 * col(*Rb, a1) + k2 * col(*Rb, a2);
    for (int i = 0; i < 3; i++) point[cnum * 3 + i] = pt[i];
    dep[cnum] = Sa[codeN] - dot(normal2, pt);
    if (dep[cnum] >= 0) {
      ret[cnum * 2] = ret[j * 2];
      ret[cnum * 2 + 1] = ret[j * 2 + 1];
      cnum++;
    }
  }
  if (cnum < 1) return 0;
  Vec3 otherNormal = col(*Rb, lanr);
  if (dot(otherNormal, normal) < 0) otherNormal = -otherNormal;
  Vec3 ortho1 = col(*Rb, a1), ortho2 = col(*Rb, a2);
  Vec3 centerB = pb;
  Vec3 faceCenter = centerB - Sb[lanr] * otherNormal;
  for (int j = 0; j < cnum; j++) {
    Contact c;
    c.point = mk3(point[j * 3] + pa[0], point[j * 3 + 1] + pa[1], point[j * 3 + 2] + pa[2]);
    s_t x = ret[j * 2], y = ret[j * 2 + 1];
    c.normal = -normal;
    c.depth = dep[j];
    c.edgeAClosestPoint = c.edgeAFixedPoint = c.edgeADir = c.edgeBClosestPoint = c.edgeBFixedPoint = c.edgeBDir = mk3(0, 0, 0);
    bool onEdgeX = std::fabs(x) == rect[0], onEdgeY = std::fabs(y) == rect[1];
    if (onEdgeX && onEdgeY) {
      // a corner of the reference rectangle: the reference box's vertex touches the incident face
      if (flip) { c.type = CT_FACE_VERTEX; c.point = c.point + c.depth * c.normal; }
      else { c.type = CT_VERTEX_FACE; c.point = c.point - c.depth * c.normal; }
    } else if (!onEdgeX && !onEdgeY) {
      c.type = flip ? CT_VERTEX_FACE : CT_FACE_VERTEX;
    } else {
      c.type = CT_EDGE_EDGE;
      s_t faceX = x > 0 ? rect[0] : -rect[0], faceY = y > 0 ? rect[1] : -rect[1];
      Vec3 faceCenterA = pa + Sa[codeN] * normal;
      Vec3 ortho1A = col(*Ra, code1), ortho2A = col(*Ra, code2);
      c.edgeAFixedPoint = faceCenterA + faceX * ortho1A + faceY * ortho2A;
      c.edgeADir = normalized(c.point - c.edgeAFixedPoint);
      c.edgeAClosestPoint = c.point;
      s_t incX = dot(ortho1, c.point) - dot(ortho1, centerB), incY = dot(ortho2, c.point) - dot(ortho2, centerB);
      s_t signX = incX == 0 ? 1.0 : (incX / std::fabs(incX)), signY = incY == 0 ? 1.0 : (incY / std::fabs(incY));
      Vec3 nearestB = (signX * Sb[a1]) * ortho1 + (signY * Sb[a2]) * ortho2 + faceCenter;
      s_t distX = std::fabs(std::fabs(incX) - Sb[a1]), distY = std::fabs(std::fabs(incY) - Sb[a2]);
      Vec3 otherB;
      if (distX < distY) otherB = (signX * Sb[a1]) * ortho1 + (-1 * signY * Sb[a2]) * ortho2 + faceCenter;
      else otherB = (-1 * signX * Sb[a1]) * ortho1 + (signY * Sb[a2]) * ortho2 + faceCenter;
      c.edgeBDir = normalized(nearestB - otherB);
      c.edgeBFixedPoint = nearestB;
      if (flip) {
        Vec3 t = c.edgeADir; c.edgeADir = c.edgeBDir; c.edgeBDir = t;
        t = c.edgeAFixedPoint; c.edgeAFixedPoint = c.edgeBFixedPoint; c.edgeBFixedPoint = t;
      }
    }
    out.push_back(c);
  }
  return cnum;
}

// Which pairs are tested: all i<j in insertion order, minus CollisionFilter.cpp:105-154
// (same body, both immobile, same skeleton with self-collision disabled).
// collideBoxSphere (DARTCollide.cpp:1482-1653; box = object 1) and collideSphereBox (:1655-1810; sphere = object 1):
// the sphere centre is clamped to the box, every clamped axis "locks" that face normal; the normal points from the second
// object towards the first.  A centre inside the box gives a plain FACE_VERTEX / VERTEX_FACE contact at the centre.
inline int sphereBoxPair(bool sphereFirst, s_t r, const Iso& Ts, const Vec3& half, const Iso& Tb, s_t clippingDepth,
                         std::vector<Contact>& out) {
  const s_t EPS = 1e-6;   // DART_COLLISION_EPS
  bool inside = true;
  Vec3 c0 = Ts.p;
  Vec3 p = apply(inverse(Tb), c0);
  s_t pa[3] = {p[0], p[1], p[2]};
  const s_t h[3] = {half[0], half[1], half[2]};
  Contact ct;
  ct.edgeAClosestPoint = ct.edgeAFixedPoint = ct.edgeADir = ct.edgeBClosestPoint = ct.edgeBFixedPoint = ct.edgeBDir = mk3(0, 0, 0);
  ct.sphereCenter = c0;
  ct.type = sphereFirst ? CT_SPHERE_BOX : CT_BOX_SPHERE;
  for (int k = 0; k < 3; k++) {
    if (pa[k] < -h[k]) { ct.faceNormal[k] = col(Tb.R, k); ct.faceLocked[k] = true; pa[k] = -h[k]; inside = false; }
    if (pa[k] > h[k]) { ct.faceNormal[k] = col(Tb.R, k); ct.faceLocked[k] = true; pa[k] = h[k]; inside = false; }
  }
  auto nearestSide = [&](s_t& mn) {
    mn = h[0] - std::fabs(pa[0]);
    int idx = 0;
    s_t t = h[1] - std::fabs(pa[1]);
    if (t < mn) { mn = t; idx = 1; }
    t = h[2] - std::fabs(pa[2]);
    if (t < mn) { mn = t; idx = 2; }
    return idx;
  };
  const s_t outward = sphereFirst ? 1.0 : -1.0;   // SphereBox: normal[idx] = p > 0 ? 1 : -1;  BoxSphere: the opposite
  if (inside) {
    s_t mn;
    int idx = nearestSide(mn);
    s_t nl[3] = {0, 0, 0};
    nl[idx] = (pa[idx] > 0.0 ? 1.0 : -1.0) * outward;
    s_t pen = mn + r;
    if (pen > clippingDepth) return 0;
    ct.type = sphereFirst ? CT_VERTEX_FACE : CT_FACE_VERTEX;
    ct.point = c0;
    ct.normal = Tb.R * mk3(nl[0], nl[1], nl[2]);
    ct.depth = pen;
    out.push_back(ct);
    return 1;
  }
  Vec3 contactpt = apply(Tb, mk3(pa[0], pa[1], pa[2]));
  Vec3 normal = sphereFirst ? c0 - contactpt : contactpt - c0;
  s_t mag = norm(normal);
  s_t pen = r - mag;
  if (pen > clippingDepth) return 0;
  if (pen < 0.0) return 0;
  if (mag > EPS) normal = (1.0 / mag) * normal;
  else {
    s_t mn;
    int idx = nearestSide(mn);
    s_t nl[3] = {0, 0, 0};
    nl[idx] = (pa[idx] > 0.0 ? 1.0 : -1.0) * outward;
    normal = Tb.R * mk3(nl[0], nl[1], nl[2]);
  }
  ct.point = contactpt;
  ct.normal = normal;
  ct.depth = pen;
  out.push_back(ct);
  return 1;
}

// collideSphereSphere (DARTCollide.cpp:1812-1880)
inline int sphereSphere(s_t r0in, const Iso& T0, s_t r1in, const Iso& T1, s_t clippingDepth, std::vector<Contact>& out) {
  const s_t EPS = 1e-6;
  s_t r0 = r0in, r1 = r1in, rsum = r0 + r1;
  Vec3 normal = T0.p - T1.p;
  s_t nsq = dot(normal, normal);
  if (nsq > rsum * rsum) return 0;
  r0 /= rsum; r1 /= rsum;
  Contact ct;
  ct.edgeAClosestPoint = ct.edgeAFixedPoint = ct.edgeADir = ct.edgeBClosestPoint = ct.edgeBFixedPoint = ct.edgeBDir = mk3(0, 0, 0);
  ct.type = CT_SPHERE_SPHERE;
  ct.centerA = T0.p; ct.radiusA = r0 * rsum; ct.centerB = T1.p; ct.radiusB = r1 * rsum;
  ct.point = r1 * T0.p + r0 * T1.p;
  if (nsq < EPS) {
    ct.normal = mk3(0, 0, 0);
    ct.depth = rsum;
    if (ct.depth > clippingDepth) return 0;
    out.push_back(ct);
    return 1;
  }
  s_t len = std::sqrt(nsq);
  ct.normal = (1.0 / len) * normal;
  ct.depth = rsum - len;
  if (ct.depth > clippingDepth) return 0;
  out.push_back(ct);
  return 1;
}

// dSegmentsClosestApproach (DARTCollide.cpp:301-381): parameters of the closest points of the segments pa->pb and ua->ub
inline void segmentsClosestApproach(const Vec3& pa, const Vec3& ua, const Vec3& pb, const Vec3& ub, s_t* alpha, s_t* beta) {
  Vec3 u = pb - pa, v = ub - ua, w = pa - ua;
  s_t a = dot(u, u), b = dot(u, v), c = dot(v, v), d = dot(u, w), e = dot(v, w);
  s_t D = a * c - b * b;
  s_t sN, sD = D, tN, tD = D;
  const s_t SMALL_NUM = 1e-15;
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
  *alpha = (std::fabs(sN) < SMALL_NUM ? 0.0 : sN / sD);
  *beta = (std::fabs(tN) < SMALL_NUM ? 0.0 : tN / tD);
}

// dDistPointToSegment (DARTCollide.cpp:384-410)
inline s_t distPointToSegment(const Vec3& p, const Vec3& ua, const Vec3& ub, s_t* alpha) {
  Vec3 v = ub - ua, w = p - ua;
  s_t c1 = dot(w, v);
  if (c1 <= 0) { *alpha = 0; return norm(p - ua); }
  s_t c2 = dot(v, v);
  if (c2 <= c1) { *alpha = 1; return norm(p - ub); }
  *alpha = c1 / c2;
  Vec3 Pb = ua + *alpha * v;
  return norm(p - Pb);
}

inline Contact blankContact() {
  Contact ct;
  ct.edgeAClosestPoint = ct.edgeAFixedPoint = ct.edgeADir = ct.edgeBClosestPoint = ct.edgeBFixedPoint = ct.edgeBDir = mk3(0, 0, 0);
  return ct;
}

// collideCapsuleCapsule (DARTCollide.cpp:4183-4284).  A capsule's axis is the z axis of its shape frame, `height` is the length
// of the cylinder part.  An end of a segment (alpha / beta within 1e-8 of 0 or 1) makes that side a sphere.
inline int capsuleCapsule(s_t height0, s_t radius0, const Iso& T0, s_t height1, s_t radius1, const Iso& T1, s_t clippingDepth,
                          std::vector<Contact>& out) {
  Vec3 pa = apply(T0, mk3(0, 0, 1) * -(height0 / 2)), pb = apply(T0, mk3(0, 0, 1) * (height0 / 2));
  Vec3 ua = apply(T1, mk3(0, 0, 1) * -(height1 / 2)), ub = apply(T1, mk3(0, 0, 1) * (height1 / 2));
  s_t alpha, beta;
  segmentsClosestApproach(pa, ua, pb, ub, &alpha, &beta);
  if (alpha < 0) alpha = 0;
  if (alpha > 1) alpha = 1;
  if (beta < 0) beta = 0;
  if (beta > 1) beta = 1;
  Vec3 closest0 = pa + (pb - pa) * alpha, closest1 = ua + (ub - ua) * beta;
  s_t dist = norm(closest0 - closest1), rsum = radius0 + radius1;
  if (!(dist <= rsum)) return 0;
  radius0 /= rsum; radius1 /= rsum;
  Contact ct = blankContact();
  ct.depth = rsum - dist;
  if (ct.depth > clippingDepth) return 0;
  ct.point = (closest0 * radius1) + (closest1 * radius0);
  ct.normal = normalized(closest0 - closest1);
  const s_t SPHERE_THRESHOLD = 1e-8;
  ct.radiusA = radius0 * rsum; ct.radiusB = radius1 * rsum;
  bool isSphere0 = std::fabs(alpha) < SPHERE_THRESHOLD || std::fabs(1 - alpha) < SPHERE_THRESHOLD;
  bool isSphere1 = std::fabs(beta) < SPHERE_THRESHOLD || std::fabs(1 - beta) < SPHERE_THRESHOLD;
  if (isSphere0 && isSphere1) { ct.type = CT_SPHERE_SPHERE; ct.centerA = closest0; ct.centerB = closest1; }
  else if (isSphere0) {
    ct.type = CT_SPHERE_PIPE;
    ct.sphereRadius = radius0 * rsum; ct.sphereCenter = closest0;
    ct.pipeRadius = radius1 * rsum; ct.pipeClosestPoint = closest1; ct.pipeFixedPoint = ua; ct.pipeDir = normalized(ub - ua);
  } else if (isSphere1) {
    ct.type = CT_PIPE_SPHERE;
    ct.pipeRadius = radius0 * rsum; ct.pipeClosestPoint = closest0; ct.pipeFixedPoint = pa; ct.pipeDir = normalized(pb - pa);
    ct.sphereRadius = radius1 * rsum; ct.sphereCenter = closest1;
  } else {
    ct.type = CT_PIPE_PIPE;
    ct.radiusA = radius0; ct.radiusB = radius1;   // the NORMALISED radii, as in the reference (:4265-4266)
    ct.edgeAFixedPoint = pa; ct.edgeAClosestPoint = closest0; ct.edgeADir = normalized(pb - pa);
    ct.edgeBFixedPoint = ua; ct.edgeBClosestPoint = closest1; ct.edgeBDir = normalized(ub - ua);
  }
  out.push_back(ct);
  return 1;
}

// collideSphereCapsule (DARTCollide.cpp:4286-4352, the sphere is object 1) / collideCapsuleSphere (:4354-4420, the capsule is)
inline int sphereCapsulePair(bool sphereFirst, s_t rSphere, const Iso& Ts, s_t height, s_t rCapsule, const Iso& Tc, s_t clippingDepth,
                             std::vector<Contact>& out) {
  s_t alpha;
  Vec3 center = Ts.p;
  Vec3 ua = apply(Tc, mk3(0, 0, 1) * -(height / 2)), ub = apply(Tc, mk3(0, 0, 1) * (height / 2));
  s_t dist = distPointToSegment(center, ua, ub, &alpha);
  s_t radius0 = sphereFirst ? rSphere : rCapsule, radius1 = sphereFirst ? rCapsule : rSphere;
  if (!(dist < radius0 + radius1)) return 0;
  Vec3 closest = ua + (ub - ua) * alpha;
  s_t rsum = radius0 + radius1;
  radius0 /= rsum; radius1 /= rsum;
  Contact ct = blankContact();
  ct.depth = rsum - dist;
  if (ct.depth > clippingDepth) return 0;
  const Vec3& first = sphereFirst ? center : closest;
  const Vec3& second = sphereFirst ? closest : center;
  ct.point = (first * radius1) + (second * radius0);
  ct.normal = normalized(first - second);
  const s_t SPHERE_THRESHOLD = 1e-8;
  ct.radiusA = radius0 * rsum; ct.radiusB = radius1 * rsum;
  bool isSphere1 = std::fabs(alpha) < SPHERE_THRESHOLD || std::fabs(1 - alpha) < SPHERE_THRESHOLD;
  if (isSphere1) { ct.type = CT_SPHERE_SPHERE; ct.centerA = first; ct.centerB = second; }
  else {
    ct.type = sphereFirst ? CT_SPHERE_PIPE : CT_PIPE_SPHERE;
    ct.sphereRadius = (sphereFirst ? radius0 : radius1) * rsum; ct.pipeRadius = (sphereFirst ? radius1 : radius0) * rsum;
    ct.sphereCenter = center; ct.pipeClosestPoint = closest; ct.pipeFixedPoint = ua; ct.pipeDir = normalized(ub - ua);
  }
  out.push_back(ct);
  return 1;
}

inline int skeletonRoot(const Model& m, int body) {
  while (body >= 0 && m.bodies[body].parent >= 0) body = m.bodies[body].parent;
  return body;
}

// two colliders are tested against each other (BodyNodeCollisionFilter::ignoresCollision, CollisionFilter.cpp:105-154)
inline bool pairIsTested(const Model& m, int i, int j) {
  const BoxCollider &bi = m.boxes[i], &bj = m.boxes[j];
  if (bi.body == bj.body) return false;
  if (bi.body < 0 && bj.body < 0) return false;
  if (bi.body >= 0 && bj.body >= 0 && m.skeleton[bi.body] == m.skeleton[bj.body]) {
    // same skeleton (CollisionFilter.cpp:138-148): only with the self-collision check on, and without the adjacent-body check not between
    // a body and its parent
    const int fi = m.selfCollision[bi.body], fj = m.selfCollision[bj.body];
    if (!((fi & 1) && (fj & 1))) return false;
    // areAdjacentBodies compares the BodyNodes' parents (CollisionFilter.cpp:150-154): the nodes of the caller's description when given
    const bool adjacent = (bi.node != -2 && bj.node != -2) ? (bi.nodeParent == bj.node || bj.nodeParent == bi.node)
                                                           : (m.bodies[bi.body].parent == bj.body || m.bodies[bj.body].parent == bi.body);
    if (!((fi & 2) && (fj & 2)) && adjacent) return false;
  }
  return true;
}

// Distinct narrow-phase points per world the DEVICE's duplicate filter remembers (model_dev.hpp SEEN_POINTS = 2 x the contact slots of the
// instantiation of the library the model runs on: 8; 16 for a model with max_contacts > 8, more than 16 colliders or more than 32
// collider pairs; 64 - the general instantiation - beyond 16 contacts, 32 colliders or 64 pairs - nimble_amd_dispatch.cpp)
inline int deviceSeenPoints(const Model& m) {
  const int nbx = (int)m.boxes.size();
  int pairs = 0;
  for (int i = 0; i + 1 < nbx; i++)
    for (int j = i + 1; j < nbx; j++) pairs += pairIsTested(m, i, j) ? 1 : 0;
  if (m.maxContacts > 64) return 256;                                     // (the 384-row general build)
  if (m.maxContacts > 16 || nbx > 32 || pairs > 64) return 128;
  return (m.maxContacts > 8 || nbx > 16 || pairs > 32) ? 32 : 16;
}

// seenListOverflow (optional): set when a point that would pass the constraint solver's filters is dropped as the duplicate of a point
// the DEVICE's duplicate filter could no longer remember (deviceSeenPoints) - there the device keeps a contact the reference does not
// have and flags the world (NBL_ST_CONTACT_OVERFLOW).
inline void collideAll(const Model& m, const std::vector<Kin>& kin, std::vector<Contact>& contacts, bool* seenListOverflow = nullptr) {
  contacts.clear();
  const int nbx = (int)m.boxes.size();
  const size_t seenCap = seenListOverflow ? (size_t)deviceSeenPoints(m) : 0;
  for (int i = 0; i + 1 < nbx; i++)
    for (int j = i + 1; j < nbx; j++) {
      const BoxCollider &bi = m.boxes[i], &bj = m.boxes[j];
      if (!pairIsTested(m, i, j)) continue;
      Iso Ti = bi.body >= 0 ? kin[bi.body].Tworld * bi.T : bi.T;
      Iso Tj = bj.body >= 0 ? kin[bj.body].Tworld * bj.T : bj.T;
      std::vector<Contact> pair;
      // dispatch on the two shape types (collide(), DARTCollide.cpp:5030-5260)
      const bool si = bi.shape == NBL_SHAPE_SPHERE, sj = bj.shape == NBL_SHAPE_SPHERE;
      const bool ci = bi.shape == NBL_SHAPE_CAPSULE, cj = bj.shape == NBL_SHAPE_CAPSULE;   // size = (radius, height, -)
      if (ci && cj) capsuleCapsule(bi.size[1], bi.size[0], Ti, bj.size[1], bj.size[0], Tj, m.clippingDepth, pair);
      else if (si && cj) sphereCapsulePair(true, bi.size[0], Ti, bj.size[1], bj.size[0], Tj, m.clippingDepth, pair);
      else if (ci && sj) sphereCapsulePair(false, bj.size[0], Tj, bi.size[1], bi.size[0], Ti, m.clippingDepth, pair);
      else if (ci || cj) continue;   // capsule-box runs libccd's MPR in the reference (DARTCollide.cpp:4422-4645): such pairs are refused at model creation
      else if (si && sj) sphereSphere(bi.size[0], Ti, bj.size[0], Tj, m.clippingDepth, pair);
      else if (si) sphereBoxPair(true, bi.size[0], Ti, 0.5 * bj.size, Tj, m.clippingDepth, pair);
      else if (sj) sphereBoxPair(false, bj.size[0], Tj, 0.5 * bi.size, Ti, m.clippingDepth, pair);
      else boxBox(Ti, 0.5 * bi.size, Tj, 0.5 * bj.size, m.clippingDepth, pair);
      // postProcess: drop points closer than 3e-12 to an already accepted contact
      for (Contact& c : pair) {
        bool close = false;
        for (size_t ti = 0; ti < contacts.size(); ti++)
          if (norm(c.point - contacts[ti].point) < 3.0e-12) {
            close = true;
            if (seenListOverflow && ti >= seenCap && dot(c.normal, c.normal) >= 1e-12 && c.depth >= 0.0 && c.depth <= m.clippingDepth) *seenListOverflow = true;
            break;
          }
        if (close) continue;
        c.boxA = i; c.boxB = j; c.bodyA = bi.body; c.bodyB = bj.body;
        contacts.push_back(c);
      }
    }
}

}  // namespace nbo
