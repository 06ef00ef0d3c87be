// Exports the reference's own popcount primitive through a C ABI so that the
// oracle's restatement can be checked against it.  This translation unit holds
// no reference code: it includes sparsestereo/hammingdistance.h from the
// reference tree (found through -I at build time) and is linked with
// sparsestereo/hammingdistance.cpp compiled from where it lies.  Outputs go to
// oracle/_ref/ only.  TEST INFRASTRUCTURE ONLY.
#include <sparsestereo/hammingdistance.h>

extern "C" __attribute__((visibility("default")))
unsigned ref_hamming32(unsigned a, unsigned b) {
    static sparsestereo::HammingDistance h;   // hammingdistance.cpp:12-23 fills the LUT
    return h.calculate(a, b);                 // hammingdistance.h:18-20
}

extern "C" __attribute__((visibility("default")))
unsigned ref_countbits16(unsigned short v) {
    static sparsestereo::HammingDistance h;
    return h.countBits(v);                    // hammingdistance.h:44-46
}
