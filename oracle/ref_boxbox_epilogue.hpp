// oracle/ref_boxbox_epilogue.hpp - TEST INFRASTRUCTURE.  C entry point over the reference's collideBoxBox compiled above (see
// ref_boxbox_prelude.hpp): T0 / T1 = 12 doubles each (R row-major, then p), full side lengths; out = up to `cap` contacts of 22 doubles:
// point(3) normal(3) depth type edgeAFixed(3) edgeADir(3) edgeBFixed(3) edgeBDir(3) pad(2).  Returns the number of contacts.
}  // namespace collision
}  // namespace dart

// out: 30 doubles per contact: point(3) normal(3) depth type sphereCenter(3) face1Normal(3) face2Normal(3) face3Normal(3) locked(3) centerA(3) centerB(3) radiusA radiusB... see below
static int packSphere(const dart::collision::CollisionResult& res, double* out, int cap) {
  int n = 0;
  for (const dart::collision::Contact& c : res.contacts) {
    if (n >= cap) break;
    double* o = out + 32 * n;
    for (int k = 0; k < 3; k++) {
      o[k] = c.point[k]; o[3 + k] = c.normal[k]; o[8 + k] = c.sphereCenter[k]; o[11 + k] = c.face1Normal[k]; o[14 + k] = c.face2Normal[k];
      o[17 + k] = c.face3Normal[k]; o[23 + k] = c.centerA[k]; o[26 + k] = c.centerB[k];
    }
    o[6] = c.penetrationDepth; o[7] = (double)c.type; o[20] = c.face1Locked; o[21] = c.face2Locked; o[22] = c.face3Locked; o[29] = c.radiusA; o[30] = c.radiusB; o[31] = 0;
    n++;
  }
  return (int)res.contacts.size();
}
static Eigen::Isometry3s isoOf(const double* T) {
  Eigen::Isometry3s A;
  for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) A(r, c) = T[3 * r + c]; A(r, 3) = T[9 + r]; }
  return A;
}
// which: 0 = collideBoxSphere(box = object 1), 1 = collideSphereBox(sphere = object 1), 2 = collideSphereSphere(radius in size[0])
extern "C" int ref_collide_sphere(int which, const double* size0, const double* T0, const double* size1, const double* T1, double clippingDepth,
                                  double* out, int cap) {
  using namespace dart::collision;
  CollisionObject o1, o2;
  CollisionOption opt;
  opt.contactClippingDepth = clippingDepth;
  CollisionResult res;
  if (which == 0) collideBoxSphere(&o1, &o2, Eigen::Vector3s(size0[0], size0[1], size0[2]), isoOf(T0), size1[0], isoOf(T1), opt, res, BOTH);
  else if (which == 1) collideSphereBox(&o1, &o2, size0[0], isoOf(T0), Eigen::Vector3s(size1[0], size1[1], size1[2]), isoOf(T1), opt, res, BOTH);
  else collideSphereSphere(&o1, &o2, size0[0], isoOf(T0), size1[0], isoOf(T1), opt, res, BOTH, BOTH);
  return packSphere(res, out, cap);
}

// which: 0 = collideCapsuleCapsule, 1 = collideSphereCapsule (sphere = object 1), 2 = collideCapsuleSphere; size = (radius, height)
// out: 48 doubles per contact: point(3) normal(3) depth type | centerA(3) centerB(3) radiusA radiusB | sphereCenter(3) pipeDir(3)
// pipeFixedPoint(3) pipeClosestPoint(3) sphereRadius pipeRadius | edgeAFixed(3) edgeADir(3) edgeBFixed(3) edgeBDir(3) edgeAClosest(3) edgeBClosest(3)
extern "C" int ref_collide_capsule(int which, const double* size0, const double* T0, const double* size1, const double* T1, double clippingDepth,
                                   double* out, int cap) {
  using namespace dart::collision;
  CollisionObject o1, o2;
  CollisionOption opt;
  opt.contactClippingDepth = clippingDepth;
  CollisionResult res;
  if (which == 0) collideCapsuleCapsule(&o1, &o2, size0[1], size0[0], isoOf(T0), size1[1], size1[0], isoOf(T1), opt, res);
  else if (which == 1) collideSphereCapsule(&o1, &o2, size0[0], isoOf(T0), size1[1], size1[0], isoOf(T1), opt, res);
  else collideCapsuleSphere(&o1, &o2, size0[1], size0[0], isoOf(T0), size1[0], isoOf(T1), opt, res);
  int n = 0;
  for (const Contact& c : res.contacts) {
    if (n >= cap) break;
    double* o = out + 48 * n;
    for (int k = 0; k < 3; k++) {
      o[k] = c.point[k]; o[3 + k] = c.normal[k]; o[8 + k] = c.centerA[k]; o[11 + k] = c.centerB[k];
      o[16 + k] = c.sphereCenter[k]; o[19 + k] = c.pipeDir[k]; o[22 + k] = c.pipeFixedPoint[k]; o[25 + k] = c.pipeClosestPoint[k];
      o[30 + k] = c.edgeAFixedPoint[k]; o[33 + k] = c.edgeADir[k]; o[36 + k] = c.edgeBFixedPoint[k]; o[39 + k] = c.edgeBDir[k];
      o[42 + k] = c.edgeAClosestPoint[k]; o[45 + k] = c.edgeBClosestPoint[k];
    }
    o[6] = c.penetrationDepth; o[7] = (double)c.type; o[14] = c.radiusA; o[15] = c.radiusB; o[28] = c.sphereRadius; o[29] = c.pipeRadius;
    n++;
  }
  return (int)res.contacts.size();
}

extern "C" int ref_collide_box_box(const double* size0, const double* T0, const double* size1, const double* T1, double clippingDepth,
                                   double* out, int cap) {
  using namespace dart::collision;
  Eigen::Isometry3s A, B;
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) { A(r, c) = T0[3 * r + c]; B(r, c) = T1[3 * r + c]; }
    A(r, 3) = T0[9 + r]; B(r, 3) = T1[9 + r];
  }
  CollisionObject o1, o2;
  CollisionOption opt;
  opt.contactClippingDepth = clippingDepth;
  CollisionResult res;
  collideBoxBox(&o1, &o2, Eigen::Vector3s(size0[0], size0[1], size0[2]), A, Eigen::Vector3s(size1[0], size1[1], size1[2]), B, opt, res);
  int n = 0;
  for (const Contact& c : res.contacts) {
    if (n >= cap) break;
    double* o = out + 22 * n;
    for (int k = 0; k < 3; k++) {
      o[k] = c.point[k]; o[3 + k] = c.normal[k]; o[8 + k] = c.edgeAFixedPoint[k]; o[11 + k] = c.edgeADir[k];
      o[14 + k] = c.edgeBFixedPoint[k]; o[17 + k] = c.edgeBDir[k];
    }
    o[6] = c.penetrationDepth; o[7] = (double)c.type; o[20] = 0; o[21] = 0;
    n++;
  }
  return (int)res.contacts.size();
}
