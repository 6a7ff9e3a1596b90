// How long does a chain of five dependent tiny launches take from the first launch call to the host seeing the last
// kernel's flag - launched one by one, as a hipGraph with constant parameters (the varying word read from mapped host
// memory), and as a hipGraph whose first node gets new parameters before every launch?  Decides whether capturing a pass of
// the coarse pyramid levels in a graph would shorten it.   hipcc --offload-arch=gfx950 -O2 -o build/ubench_launch_chain tools/ubench_launch_chain.hip
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void first_kernel(uint32_t seq_arg, const uint32_t* seq_mapped, uint32_t* dev) {
  if (threadIdx.x == 0) dev[0] = seq_mapped ? __hip_atomic_load(seq_mapped, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : seq_arg;
}
__global__ void middle_kernel(uint32_t* dev, int k) {
  if (threadIdx.x == 0) dev[k] = dev[k - 1];
}
__global__ void last_kernel(uint32_t* dev, int k, uint32_t* ready) {
  if (threadIdx.x == 0) __hip_atomic_store(ready, dev[k - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  uint32_t *dev, *ready, *ready_m, *seq, *seq_m;
  CK(hipMalloc(reinterpret_cast<void**>(&dev), 64));
  CK(hipHostMalloc(reinterpret_cast<void**>(&ready), 64, hipHostMallocMapped | hipHostMallocCoherent));
  CK(hipHostGetDevicePointer(reinterpret_cast<void**>(&ready_m), ready, 0));
  CK(hipHostMalloc(reinterpret_cast<void**>(&seq), 64, hipHostMallocMapped | hipHostMallocCoherent));
  CK(hipHostGetDevicePointer(reinterpret_cast<void**>(&seq_m), seq, 0));
  *ready = 0;
  const int reps = 2000;
  auto wait = [&](uint32_t v) { while (__atomic_load_n(ready, __ATOMIC_ACQUIRE) != v) {} };
  auto report = [&](const char* what, std::vector<double>& t) {
    std::sort(t.begin(), t.end());
    std::printf("%-44s median %.2f us  p10 %.2f  p90 %.2f\n", what, 1e6 * t[t.size() / 2], 1e6 * t[t.size() / 10], 1e6 * t[t.size() * 9 / 10]);
  };
  uint32_t s = 0;
  {  // one by one
    std::vector<double> t;
    for (int r = 0; r < reps; ++r) {
      ++s;
      const double t0 = now();
      hipLaunchKernelGGL(first_kernel, dim3(1), dim3(64), 0, st, s, nullptr, dev);
      for (int k = 1; k < 4; ++k) hipLaunchKernelGGL(middle_kernel, dim3(1), dim3(64), 0, st, dev, k);
      hipLaunchKernelGGL(last_kernel, dim3(1), dim3(64), 0, st, dev, 4, ready_m);
      const double t1 = now();
      wait(s);
      if (r >= 100) t.push_back(now() - t0);
      (void)t1;
    }
    report("5 launches, one by one", t);
  }
  {  // one launch only (floor)
    std::vector<double> t;
    for (int r = 0; r < reps; ++r) {
      ++s;
      const double t0 = now();
      hipLaunchKernelGGL(first_kernel, dim3(1), dim3(64), 0, st, s, nullptr, dev);
      hipLaunchKernelGGL(last_kernel, dim3(1), dim3(64), 0, st, dev, 1, ready_m);
      wait(s);
      if (r >= 100) t.push_back(now() - t0);
    }
    report("2 launches", t);
  }
  hipGraph_t graph;
  hipGraphExec_t exec;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  hipLaunchKernelGGL(first_kernel, dim3(1), dim3(64), 0, st, 0u, seq_m, dev);
  for (int k = 1; k < 4; ++k) hipLaunchKernelGGL(middle_kernel, dim3(1), dim3(64), 0, st, dev, k);
  hipLaunchKernelGGL(last_kernel, dim3(1), dim3(64), 0, st, dev, 4, ready_m);
  CK(hipStreamEndCapture(st, &graph));
  CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
  {
    std::vector<double> t;
    for (int r = 0; r < reps; ++r) {
      ++s;
      const double t0 = now();
      __atomic_store_n(seq, s, __ATOMIC_RELEASE);
      CK(hipGraphLaunch(exec, st));
      wait(s);
      if (r >= 100) t.push_back(now() - t0);
    }
    report("graph of 5, constant parameters", t);
  }
  {
    size_t n = 0;
    CK(hipGraphGetNodes(graph, nullptr, &n));
    std::vector<hipGraphNode_t> nodes(n);
    CK(hipGraphGetNodes(graph, nodes.data(), &n));
    hipGraphNode_t first = nullptr;
    for (auto nd : nodes) {
      hipKernelNodeParams p{};
      if (hipGraphKernelNodeGetParams(nd, &p) == hipSuccess && p.func == reinterpret_cast<void*>(first_kernel)) first = nd;
    }
    if (!first) { std::printf("first node not found among %zu\n", n); return 1; }
    std::vector<double> t;
    for (int r = 0; r < reps; ++r) {
      ++s;
      const double t0 = now();
      uint32_t a0 = s; const uint32_t* a1 = nullptr; uint32_t* a2 = dev;
      void* args[3] = {&a0, &a1, &a2};
      hipKernelNodeParams p{};
      p.func = reinterpret_cast<void*>(first_kernel);
      p.gridDim = dim3(1); p.blockDim = dim3(64); p.sharedMemBytes = 0; p.kernelParams = args; p.extra = nullptr;
      CK(hipGraphExecKernelNodeSetParams(exec, first, &p));
      CK(hipGraphLaunch(exec, st));
      wait(s);
      if (r >= 100) t.push_back(now() - t0);
    }
    report("graph of 5, first node re-parametrised", t);
  }
  CK(hipStreamSynchronize(st));
  return 0;
}
