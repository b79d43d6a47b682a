// oracle/ref_geometry_shells.hpp - TEST INFRASTRUCTURE.  Class shells for the member functions oracle/ref_build.py reads from the reference
// after the free functions of Geometry.cpp: the declarations those definitions need, nothing else (own code; the bodies come from
// /root/reference at build time).
//   dart::dynamics::FreeJoint          convertToPositions, convertToTransform, integratePositionsExplicit (FreeJoint.cpp:65-81, 922-947)
//   dart::constraint::ContactConstraint getTangentBasisMatrixODE, getTangentBasisMatrixODEGradient (ContactConstraint.cpp:734-876)
#define DART_USE_IDENTITY_JACOBIAN     // the build the reference ships (dart/CMakeLists.txt: add_definitions(-DDART_USE_IDENTITY_JACOBIAN))
namespace dart {
namespace math {
namespace suffixes {}
}  // namespace math
namespace dynamics {
class FreeJoint {
public:
  static Eigen::Vector6s convertToPositions(const Eigen::Isometry3s& _tf);
  static Eigen::Isometry3s convertToTransform(const Eigen::Vector6s& _positions);
  Eigen::VectorXs integratePositionsExplicit(const Eigen::VectorXs& pos, const Eigen::VectorXs& vel, s_t dt);
};
}  // namespace dynamics
namespace constraint {
class ContactConstraint {
public:
  typedef Eigen::Mat<3, 2> TangentBasisMatrix;
  TangentBasisMatrix getTangentBasisMatrixODE(const Eigen::Vector3s& n);
  TangentBasisMatrix getTangentBasisMatrixODEGradient(const Eigen::Vector3s& n, const Eigen::Vector3s& g);
  Eigen::Vector3s mFirstFrictionalDirection = Eigen::Vector3s::UnitZ();    // ContactConstraint.cpp:81
};
}  // namespace constraint
}  // namespace dart
