// Host-only check of CreatePlan(N, tuner_file) in include/tensor_fft.hpp (the reference's overload, Plan.h:197-255,
// plus the sixth column tools/tuner.py appends): usage tuner_file_host FILE N -> prints "ok <variant>" or "refused".
// Touches no device, so tests/test_capi_host.py can run it without a GPU.
#include <cstdlib>
#include <iostream>

#include "tensor_fft.hpp"

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  const int n = std::atoi(argv[2]);
  auto plan = CreatePlan(n, std::string(argv[1]));
  if (!plan) {
    std::cout << "refused" << std::endl;
    return 0;
  }
  std::cout << "ok " << plan->tfft_variant_ << " " << plan->amount_of_r16_steps_ << " " << plan->amount_of_r2_steps_ << " "
            << static_cast<int>(plan->base_fft_mode_);
  // every (batch, variant, launch_iters) line of this length, then what a batch of argv[3] would run with
  for (const auto& t : plan->tfft_tuned_) std::cout << " [" << t.batch << " " << t.variant << " " << t.launch_iters << "]";
  if (argc > 3) {
    const auto pick = tfft_detail::tuned_for_batch(*plan, static_cast<uint64_t>(std::atoll(argv[3])));
    std::cout << " pick " << pick.first << " " << pick.second;
  }
  std::cout << std::endl;
  return 0;
}
