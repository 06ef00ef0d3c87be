// Exports the reference's own linear solver through a C ABI so that the oracle's plane fit can be
// checked against it.  This translation unit holds no reference code: it includes
// linear-system-solving/CramerTheorem.h from the reference tree (found through -I at build time) and is
// linked with CramerTheorem.cpp and Matrix.cpp compiled from where they lie.  Outputs go to
// oracle/_ref/ only.  TEST INFRASTRUCTURE ONLY.
#include <vector>
#include <linear-system-solving/CramerTheorem.h>

// xyd: three rows (x, y, d).  abc: the plane through them narrowed to float exactly as the reference's
// Plane constructor does (plane/Plane.cpp:27-65): rows (x, y, 1 | d), Cramer in double, float members;
// a singular system leaves (0,0,0).
extern "C" __attribute__((visibility("default")))
void ref_plane_fit(const double *xyd, float *abc) {
    std::vector<std::vector<double>> coefficients;
    for (int i = 0; i < 3; ++i) coefficients.push_back({xyd[3 * i], xyd[3 * i + 1], 1.0, xyd[3 * i + 2]});
    CramerTheorem calculator(3, coefficients);
    calculator.calculate();
    abc[0] = abc[1] = abc[2] = 0.0f;
    if (calculator.getResult() == CramerResult::FINITE_RESULTS) {
        std::vector<double> r = calculator.getFiniteResults();
        abc[0] = r[0]; abc[1] = r[1]; abc[2] = r[2];
    }
}
