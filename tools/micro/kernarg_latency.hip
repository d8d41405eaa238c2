// Dependent-launch cost of a trivial kernel as a function of the size of its by-value argument (the library passes its ~2 KB
// device-mirror struct DM by value to every kernel) against the same data behind a pointer to a device-resident copy.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
template <int N> struct Big { double *p[N]; };
template <int N> __global__ void k_val(Big<N> b, int i) { if (threadIdx.x == 0 && blockIdx.x == 0) b.p[i][0] += 1.0; }
template <int N> __global__ void k_ptr(const Big<N> *__restrict__ b, int i) { if (threadIdx.x == 0 && blockIdx.x == 0) b->p[i][0] += 1.0; }
template <int N> void run(const char *name) {
  double *d; hipMalloc(&d, 64); hipMemset(d, 0, 64);
  Big<N> h; for (int i = 0; i < N; i++) h.p[i] = d;
  Big<N> *dev; hipMalloc(&dev, sizeof(h)); hipMemcpy(dev, &h, sizeof(h), hipMemcpyHostToDevice);
  hipStream_t s; hipStreamCreate(&s);
  const int R = 4000, G = 800;   // 800 blocks of 256 threads ~ the pi column kernels
  for (int mode = 0; mode < 2; mode++) {
    for (int w = 0; w < 2; w++) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      auto t0 = std::chrono::steady_clock::now();
      hipEventRecord(e0, s);
      for (int r = 0; r < R; r++) {
        if (mode == 0) hipLaunchKernelGGL(k_val<N>, dim3(G), dim3(256), 0, s, h, r % N);
        else hipLaunchKernelGGL(k_ptr<N>, dim3(G), dim3(256), 0, s, dev, r % N);
      }
      hipEventRecord(e1, s);
      double host_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / R;
      hipStreamSynchronize(s);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (w == 1) printf("%s args %5zu B  %s: %.2f us per dependent launch on the GPU, %.2f us host enqueue\n", name, sizeof(h), mode ? "pointer " : "by value", ms * 1e3 / R, host_us);
    }
  }
}
int main() { run<1>("1"); run<64>("64"); run<256>("256"); run<320>("320"); return 0; }
