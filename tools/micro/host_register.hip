// How should a 288 MB pageable host image reach the GPU?  (a) pin it in place (hipHostRegister) and DMA,
// (b) plain hipMemcpy from pageable memory, (c) copy through a pinned staging ring with 1 / 4 host threads.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
  const size_t bytes = 4000ull * 6000 * 12;
  char* host = (char*)malloc(bytes);
  memset(host, 1, bytes);
  void* dev;
  (void)hipMalloc(&dev, bytes);
  hipStream_t st;
  (void)hipStreamCreate(&st);
  for (int rep = 0; rep < 3; ++rep) {
    double t0 = now();
    hipError_t e = hipHostRegister(host, bytes, hipHostRegisterDefault);
    double t1 = now();
    (void)hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, st);
    (void)hipStreamSynchronize(st);
    double t2 = now();
    (void)hipHostUnregister(host);
    double t3 = now();
    printf("register %s: pin %.2f ms, copy %.2f ms (%.1f GB/s), unpin %.2f ms\n", hipGetErrorString(e), (t1 - t0) * 1e3,
           (t2 - t1) * 1e3, bytes / (t2 - t1) / 1e9, (t3 - t2) * 1e3);
  }
  for (int rep = 0; rep < 3; ++rep) {
    double t0 = now();
    (void)hipMemcpy(dev, host, bytes, hipMemcpyHostToDevice);
    double t1 = now();
    printf("pageable hipMemcpy: %.2f ms (%.1f GB/s)\n", (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9);
  }
  const size_t chunk = 32u << 20;
  void* pin[2];
  hipEvent_t freed[2];
  for (int i = 0; i < 2; ++i) {
    (void)hipHostMalloc(&pin[i], chunk, hipHostMallocDefault);
    (void)hipEventCreateWithFlags(&freed[i], hipEventDisableTiming);
  }
  for (int threads : {1, 2, 4, 8}) {
    for (int rep = 0; rep < 2; ++rep) {
      double t0 = now();
      int b = 0;
      for (size_t off = 0; off < bytes; off += chunk, b ^= 1) {
        const size_t n = bytes - off < chunk ? bytes - off : chunk;
        (void)hipEventSynchronize(freed[b]);
        std::vector<std::thread> pool;
        const size_t part = (n + threads - 1) / threads;
        for (int t = 1; t < threads; ++t) {
          const size_t o = t * part;
          if (o < n) pool.emplace_back([=] { memcpy((char*)pin[b] + o, host + off + o, n - o < part ? n - o : part); });
        }
        memcpy(pin[b], host + off, part < n ? part : n);
        for (auto& th : pool) th.join();
        (void)hipMemcpyAsync((char*)dev + off, pin[b], n, hipMemcpyHostToDevice, st);
        (void)hipEventRecord(freed[b], st);
      }
      (void)hipStreamSynchronize(st);
      double t1 = now();
      if (rep) printf("staged, %d thread(s): %.2f ms (%.1f GB/s)\n", threads, (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9);
    }
  }
  return 0;
}
