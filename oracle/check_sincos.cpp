// check_sincos.cpp -- TEST INFRASTRUCTURE.  Exhaustive comparison of the canonical det_sincos() (App. C-3)
// with this image's glibc cosf()/sinf() -- what the reference's `(float)cos(angle)` binds to
// (ORBextractor.cc:113 with `using namespace std`, :67) -- over EVERY float in [0, 2*pi + margin].
// Usage: ./check_sincos          (prints mismatch counts; takes ~10-20 s on 8 cores)
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
extern "C" void orc_det_sincos(float, float*, float*);
#include "orb_oracle.cpp"
int main() {
    float hi = 6.2831855f * 1.0001f;
    uint32_t hb; std::memcpy(&hb, &hi, 4);
    long long nc = 0, ns = 0, n = 0; int maxulp = 0;
    #pragma omp parallel for reduction(+ : nc, ns, n) reduction(max : maxulp) schedule(static, 1 << 16)
    for (uint32_t b = 0; b <= hb; b++) {
        float x; std::memcpy(&x, &b, 4);
        float c, s; orc_det_sincos(x, &c, &s);
        float rc = cosf(x), rs = sinf(x);
        nc += std::memcmp(&c, &rc, 4) != 0;
        ns += std::memcmp(&s, &rs, 4) != 0;
        int32_t ic, irc, is, irs; std::memcpy(&ic,&c,4); std::memcpy(&irc,&rc,4); std::memcpy(&is,&s,4); std::memcpy(&irs,&rs,4);
        if ((ic ^ irc) >= 0) maxulp = std::max(maxulp, std::abs(ic - irc));
        if ((is ^ irs) >= 0) maxulp = std::max(maxulp, std::abs(is - irs));
        n++;
    }
    std::printf("floats checked: %lld  cos mismatches vs glibc cosf: %lld  sin mismatches vs glibc sinf: %lld  max ulp distance (same sign): %d\n", n, nc, ns, maxulp);
    return 0;
}
