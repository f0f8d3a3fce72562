// tools/d2h_probe.cpp -- what limits the download of the host-buffer path: D2H rates of pinned blocks by size, flag and
// issuing thread.  hipcc -O2 tools/d2h_probe.cpp -o exp_libs/d2h_probe -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
__global__ void busy_kernel(float* x, int iters) {
  float v = x[threadIdx.x];
  for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
  x[threadIdx.x + blockIdx.x * blockDim.x] = v;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void run(const char* name, unsigned flags, size_t bytes, int pieces, bool other_thread) {
  char* d = nullptr;
  hipMalloc((void**)&d, bytes * pieces);
  std::vector<char*> h(3);
  for (auto& p : h) hipHostMalloc((void**)&p, bytes, flags);
  hipStream_t s;
  hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  auto body = [&] {
    hipSetDevice(0);
    for (int rep = 0; rep < 2; ++rep) {
      const double t0 = now();
      for (int i = 0; i < pieces; ++i) hipMemcpyAsync(h[i % 3], d + (size_t)i * bytes, bytes, hipMemcpyDeviceToHost, s);
      hipStreamSynchronize(s);
      const double dt = now() - t0;
      if (rep == 1) std::printf("%-34s %6.1f MB x %2d  %s: %6.1f GB/s\n", name, bytes / 1e6, pieces, other_thread ? "thread" : "main  ", bytes * pieces / dt / 1e9);
    }
  };
  if (other_thread) std::thread(body).join(); else body();
  for (auto p : h) hipHostFree(p);
  hipFree(d);
  hipStreamDestroy(s);
}
int main() {
  for (size_t mb : {8, 32, 64})
    for (int th = 0; th < 2; ++th) {
      run("hipHostMallocDefault", hipHostMallocDefault, mb << 20, 8, th);
      run("hipHostMallocPortable", hipHostMallocPortable, mb << 20, 8, th);
      run("hipHostMallocNumaUser", hipHostMallocNumaUser, mb << 20, 8, th);
      run("hipHostMallocNonCoherent", hipHostMallocNonCoherent, mb << 20, 8, th);
    }
  // D2H while kernels run on another stream (all CUs busy / a quarter of them)
  for (int blocks : {256 * 8, 64}) {
    size_t bytes = 64 << 20; char *d, *h; float* x; hipMalloc((void**)&d, bytes); hipHostMalloc((void**)&h, bytes, 0);
    hipMalloc((void**)&x, (size_t)blocks * 256 * 4 + 1024);
    hipStream_t s, k; hipStreamCreateWithFlags(&s, hipStreamNonBlocking); hipStreamCreateWithFlags(&k, hipStreamNonBlocking);
    for (int rep = 0; rep < 2; ++rep) {
      for (int j = 0; j < 40; ++j) hipLaunchKernelGGL(busy_kernel, dim3(blocks), dim3(256), 0, k, x, 200000);
      double t0 = now();
      for (int i = 0; i < 8; ++i) hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, s);
      hipStreamSynchronize(s);
      double dt = now() - t0;
      hipStreamSynchronize(k);
      if (rep) std::printf("D2H 64 MB x 8 beside %d busy blocks: %.1f GB/s (kernels took %.1f ms more)\n", blocks, bytes * 8 / dt / 1e9, (now() - t0 - dt) * 1e3);
    }
  }
  // H2D for comparison
  {
    size_t bytes = 32 << 20; char *d, *h; hipMalloc((void**)&d, bytes); hipHostMalloc((void**)&h, bytes, 0);
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    for (int rep = 0; rep < 2; ++rep) { double t0 = now(); for (int i = 0; i < 8; ++i) hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); if (rep) std::printf("H2D default 32 MB x 8: %.1f GB/s\n", bytes * 8 / (now() - t0) / 1e9); }
  }
  return 0;
}
