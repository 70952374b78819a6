// Exhaustive host-side check of gaast_amd/csrc/device/spinor_basis.hpp (no GPU): for every (alpha, lambda)
// of m = 3..6 the chosen basis is invertible, puts lambda on one coordinate and alpha on the top one or
// nowhere, and keeps the pairing c.z (L and L^-T are inverse transposes).
#include <cstdio>
#include <cstdlib>

#include "spinor_basis.hpp"

using gaast::SpinorBasis;

int main() {
    long cases = 0;
    for (int m = 3; m <= 6; ++m) {
        const uint32_t N = 1u << m;
        for (uint32_t alpha = 0; alpha < N; ++alpha)
            for (uint32_t lam = 0; lam < N; ++lam) {
                const SpinorBasis b = gaast::choose_spinor_basis(m, alpha, lam);
                // L L^-1 = 1  <=>  (L^-T)^T L ... check through the pairing instead: c.z == c'.z'
                for (uint32_t c = 0; c < N; ++c)
                    for (uint32_t z = 0; z < N; z += (m > 4 ? 5 : 1)) {
                        if (SpinorBasis::par(c & z) != SpinorBasis::par(b.map_x(c) & b.map_z(z))) {
                            std::printf("pairing broken m=%d alpha=%u lam=%u\n", m, alpha, lam);
                            return 1;
                        }
                    }
                // f = alpha.x ^ lam.z  ==  x'_top [has_alpha] ^ z'_lam_bit for every (x, z)
                for (uint32_t x = 0; x < N; ++x)
                    for (uint32_t z = 0; z < N; z += (m > 4 ? 3 : 1)) {
                        const uint32_t f = SpinorBasis::par(alpha & x) ^ SpinorBasis::par(lam & z);
                        const uint32_t x2 = b.map_x(x), z2 = b.map_z(z);
                        const uint32_t g = (b.has_alpha ? (x2 >> (m - 1)) & 1u : 0u) ^
                                           (b.lam_bit >= 0 ? (z2 >> b.lam_bit) & 1u : 0u);
                        if (f != g) {
                            std::printf("parity mismatch m=%d alpha=%u lam=%u x=%u z=%u\n", m, alpha, lam, x, z);
                            return 1;
                        }
                    }
                if ((lam == 0) != (b.lam_bit < 0) || (alpha == 0) != !b.has_alpha) {
                    std::printf("case flags wrong m=%d alpha=%u lam=%u\n", m, alpha, lam);
                    return 1;
                }
                if (b.lam_bit >= 0 && b.lam_bit != m - 1 && b.lam_bit != m - 2) return 1;
                ++cases;
            }
    }
    std::printf("OK %ld cases\n", cases);
    return 0;
}
