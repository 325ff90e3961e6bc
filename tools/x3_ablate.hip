// Where does the time of the split-precision convolution (csrc/conv_x3.hip) go on the layers that are NOT matrix-core bound?
// Standalone (no torch): the product source compiled with -DUPF_X3_ABL=<bits> (1 no matrix phase, 2 no split / LDS stores,
// 4 no global loads, 8 matrix phase without LDS reads), timed with HIP events around 30 launches, at [8, Cin, 96, 320] fp32.
//   for a in 0 1 2 4 6 8 9 14; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -Xclang -target-feature -Xclang -packed-fp32-ops -DUPF_X3_ABL=$a \
//       -I upflow_pytorch_amd/csrc -I include tools/x3_ablate.hip upflow_pytorch_amd/csrc/api.hip -o /tmp/x3a_$a; /tmp/x3a_$a; done
#include "../upflow_pytorch_amd/csrc/conv_x3.hip"
#include <vector>
#include <cstdio>
#include <cstdlib>
#include <algorithm>

static float time_layer(int B, int Cin, int Cout, int H, int W, int d) {
  const size_t nx = (size_t)B * Cin * H * W, ny = (size_t)B * Cout * H * W, nw = (size_t)Cout * Cin * 9;
  float *x, *y, *w, *b; void* wp;
  (void)hipMalloc(&x, nx * 4); (void)hipMalloc(&y, ny * 4); (void)hipMalloc(&w, nw * 4); (void)hipMalloc(&b, Cout * 4);
  (void)hipMalloc(&wp, upf_conv_x3_packed_bytes(Cin, Cout, 3));
  std::vector<float> h(nx);
  for (auto& v : h) v = (float)(rand() % 2001 - 1000) * 1e-3f;
  (void)hipMemcpy(x, h.data(), nx * 4, hipMemcpyHostToDevice);
  std::vector<float> hw(nw);
  for (auto& v : hw) v = (float)(rand() % 2001 - 1000) * 2e-5f;
  (void)hipMemcpy(w, hw.data(), nw * 4, hipMemcpyHostToDevice);
  (void)hipMemset(b, 0, Cout * 4);
  if (upf_conv_x3_pack_weights(w, wp, Cin, Cout, 3, nullptr)) { printf("pack failed\n"); exit(1); }
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  std::vector<float> t;
  for (int rep = 0; rep < 34; ++rep) {
    (void)hipEventRecord(e0, 0);
    if (upf_conv_x3_forward(x, (long long)Cin * H * W, wp, b, y, (long long)Cout * H * W, B, Cin, Cout, H, W, 3, d, 1, 0.1f, 3, nullptr)) { printf("launch failed: %s\n", upf_last_error()); exit(1); }
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (rep >= 4) t.push_back(ms * 1e3f);
  }
  std::sort(t.begin(), t.end());
  (void)hipFree(x); (void)hipFree(y); (void)hipFree(w); (void)hipFree(b); (void)hipFree(wp);
  return t[t.size() / 2];
}

int main() {
  printf("ABL=%-2d  531->32 %7.1f us   184->3 %6.1f us   64->32 %6.1f us   565->128 %7.1f us   128->128 d2 %6.1f us\n", UPF_X3_ABL,
         time_layer(8, 531, 32, 96, 320, 1), time_layer(8, 184, 3, 96, 320, 1), time_layer(8, 64, 32, 96, 320, 1),
         time_layer(8, 565, 128, 96, 320, 1), time_layer(8, 128, 128, 96, 320, 2));
  return 0;
}
