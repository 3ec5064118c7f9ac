// Host check of k4096::Rotor (the work distribution of the persistent kernels): for every grid size and item count each item is
// taken exactly once, the look-ahead equals the next item, and a workgroup's consecutive items run through all residues mod 8.
#include <cstdint>
#include <cstdio>
#include <vector>

#include "../../tensor-fft_amd/csrc/k4096.hpp"

int main() {
  const uint32_t grids[] = {1, 2, 7, 8, 32, 255, 256, 512};
  const uint32_t totals[] = {1, 5, 8, 255, 256, 257, 1000, 16384, 16385};
  for (uint32_t grid : grids)
    for (uint32_t total : totals) {
      std::vector<int> seen(total, 0);
      for (uint32_t g = 0; g < grid; ++g) {
        k4096::Rotor rot(g, grid);
        uint32_t mask = 0, taken = 0;
        for (uint32_t it = rot.item(); it < total; rot.advance(), it = rot.item()) {
          ++seen[it];
          mask |= 1u << (it & 7);
          ++taken;
          const uint32_t ahead = rot.peek();
          k4096::Rotor next = rot;
          next.advance();
          if (ahead != next.item()) {
            std::printf("peek mismatch grid %u total %u\n", grid, total);
            return 1;
          }
        }
        if (grid % 8 == 0 && taken >= 8 && mask != 0xffu) {
          std::printf("workgroup %u of %u saw residues %02x only (total %u)\n", g, grid, mask, total);
          return 1;
        }
      }
      for (uint32_t i = 0; i < total; ++i)
        if (seen[i] != 1) {
          std::printf("item %u taken %d times (grid %u total %u)\n", i, seen[i], grid, total);
          return 1;
        }
    }
  std::printf("ok\n");
  return 0;
}
