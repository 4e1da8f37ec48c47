// tools/device_tail.hip -- could the tail of gen_dist() (reference ngsDist.cpp:372-401: d = sum / cnt, then -log(1 - d)
// or the JC69 form) run on the DEVICE between the two collectives of a multi-GPU job, instead of device -> host ->
// device?  Only if its cells carry the bits of the host's: the product prints them with "%.10f" and promises
// byte-identical output.  This tool measures it: random (sum, cnt) cells plus the special
// classes (d = 0 -> -0.0, d = 1 -> inf, JC69 saturation -> nan, cnt = 0), the tail on the device (the device
// library's log) against ngd_finish() of libngsdist_amd.so (g++ -O3 + glibc, the product's tail), bit for bit.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/device_tail.hip -o tools/device_tail -Lngsdist_amd
//         -lngsdist_amd -Wl,-rpath,'$ORIGIN/../ngsdist_amd'        (tools/device_tail.sh)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <cmath>
#include <vector>

#include "../include/ngsdist_amd.h"

__global__ void k_tail(const double *sum, const unsigned long long *cnt, uint64_t n, int model, double *out) {
  const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  double d = sum[k];
  d /= (double)cnt[k];
  if (model == 1)
    d = -log(1 - d);
  else if (model == 2)
    d = -log(1 - (d * 4 / 3)) * 3 / 4;
  out[k] = d;
}

static uint64_t mix(uint64_t z) {  // splitmix64 finaliser
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

int main(int argc, char **argv) {
  const uint64_t total = argc > 1 ? strtoull(argv[1], nullptr, 10) : 100000000ull;
  const uint64_t chunk = 10000000ull;
  std::vector<double> sum(chunk), host(chunk), dev(chunk);
  std::vector<uint64_t> cnt(chunk);
  double *d_sum, *d_out;
  unsigned long long *d_cnt;
  if (hipMalloc((void **)&d_sum, chunk * 8) != hipSuccess || hipMalloc((void **)&d_out, chunk * 8) != hipSuccess ||
      hipMalloc((void **)&d_cnt, chunk * 8) != hipSuccess) {
    fprintf(stderr, "no device memory\n");
    return 2;
  }
  for (int model = 0; model <= 2; model++) {
    uint64_t n_diff = 0, n_cells = 0, n_special_diff = 0, worst_ulp = 0, n_text = 0;
    for (uint64_t base = 0; base < total; base += chunk) {
      const uint64_t n = std::min(chunk, total - base);
      for (uint64_t k = 0; k < n; k++) {
        const uint64_t z = mix((base + k) * 0x9E3779B97F4A7C15ull + model);
        // counts as a job has them (a few sites ... a few million), sums = d * cnt with d over (0, 1): mostly small
        // distances (what real data gives), some near the saturation of the models; every 1000th cell a special class
        const uint64_t c = 1 + (z >> 40) % 3000000;
        double u = ((mix(z) >> 11) + 0.5) * 0x1p-53;
        double d = (z & 3) ? u * u * 0.6 : u;
        cnt[k] = c;
        sum[k] = d * (double)c;
        if ((base + k) % 1000 == 0) {
          switch (((base + k) / 1000) % 5) {
            case 0: sum[k] = 0.0; break;                 // identical individuals: -0.0 under models 1, 2
            case 1: sum[k] = (double)c; break;           // d = 1: inf / nan
            case 2: sum[k] = 0.75 * (double)c; break;    // JC69 saturation
            case 3: cnt[k] = 0; break;                   // no valid site: nan (0/0) or inf
            case 4: sum[k] = 0.5 * (double)(c / 2 * 2); cnt[k] = c / 2 * 2 ? c / 2 * 2 : 2; break;  // called genotypes: dyadic d
          }
        }
      }
      if (ngd_finish(sum.data(), cnt.data(), n, 0, model, host.data()) != NGD_OK) return 3;
      hipMemcpy(d_sum, sum.data(), n * 8, hipMemcpyHostToDevice);
      hipMemcpy(d_cnt, cnt.data(), n * 8, hipMemcpyHostToDevice);
      hipLaunchKernelGGL(k_tail, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, d_sum, d_cnt, n, model, d_out);
      if (hipMemcpy(dev.data(), d_out, n * 8, hipMemcpyDeviceToHost) != hipSuccess) return 4;
      for (uint64_t k = 0; k < n; k++) {
        uint64_t a, b;
        memcpy(&a, &host[k], 8);
        memcpy(&b, &dev[k], 8);
        n_cells++;
        if (a == b) continue;
        if (std::isnan(host[k]) && std::isnan(dev[k]) && (a >> 63) == (b >> 63)) continue;  // same nan / -nan in print
        n_diff++;
        if ((base + k) % 1000 == 0) n_special_diff++;
        if (std::isfinite(host[k]) && std::isfinite(dev[k])) {
          const uint64_t ulp = a > b ? a - b : b - a;
          if (ulp > worst_ulp) worst_ulp = ulp;
          char ta[64], tb[64];
          snprintf(ta, sizeof ta, "%.10f", host[k]);
          snprintf(tb, sizeof tb, "%.10f", dev[k]);
          if (strcmp(ta, tb)) n_text++;
        } else {
          n_text++;
        }
      }
    }
    printf("evol_model %d: %llu cells, %llu differ in their bits (%.3g of all; %llu of them special-class cells), largest "
           "difference %llu ulp, %llu would print differently with %%.10f\n",
           model, (unsigned long long)n_cells, (unsigned long long)n_diff, (double)n_diff / (double)n_cells,
           (unsigned long long)n_special_diff, (unsigned long long)worst_ulp, (unsigned long long)n_text);
  }
  return 0;
}
