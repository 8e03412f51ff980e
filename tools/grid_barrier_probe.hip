// How much does an in-kernel grid barrier (agent-scope release/acquire + arrival counter) cost on MI355X, compared
// with a kernel boundary?  Decides whether chaining small GEMMs in one persistent kernel can pay.
//   hipcc --offload-arch=gfx950 -O3 tools/grid_barrier_probe.hip -o /tmp/gbp && /tmp/gbp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ bool grid_barrier(unsigned* counter, unsigned target) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int spins = 0;
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1 << 22)) { ok = false; break; }   // bail out instead of hanging the GPU
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
  return ok;
}

// every round: each block writes a value that depends on what ANOTHER block wrote in the previous round
__global__ void __launch_bounds__(512) k_chain(float* buf, int n_per_block, unsigned* counter, int rounds, int* fail) {
  const int nb = gridDim.x;
  for (int r = 0; r < rounds; ++r) {
    const float* src = buf + (size_t)(r & 1) * nb * n_per_block + (size_t)((blockIdx.x + 1) % nb) * n_per_block;
    float* dst = buf + (size_t)((r + 1) & 1) * nb * n_per_block + (size_t)blockIdx.x * n_per_block;
    for (int i = threadIdx.x; i < n_per_block; i += blockDim.x) dst[i] = src[i] + 1.0f;
    if (!grid_barrier(counter, (unsigned)(r + 1) * nb)) { if (threadIdx.x == 0) *fail = 1; return; }
  }
}

__global__ void __launch_bounds__(512) k_round(float* buf, int n_per_block, int r) {
  const int nb = gridDim.x;
  const float* src = buf + (size_t)(r & 1) * nb * n_per_block + (size_t)((blockIdx.x + 1) % nb) * n_per_block;
  float* dst = buf + (size_t)((r + 1) & 1) * nb * n_per_block + (size_t)blockIdx.x * n_per_block;
  for (int i = threadIdx.x; i < n_per_block; i += blockDim.x) dst[i] = src[i] + 1.0f;
}

int main() {
  const int rounds = 200;
  for (int nb : {16, 64, 256}) {
    for (int npb : {512, 8192}) {
      float* buf; unsigned* counter; int* fail;
      CK(hipMalloc(&buf, sizeof(float) * 2 * nb * npb));
      CK(hipMalloc(&counter, 4)); CK(hipMalloc(&fail, 4));
      CK(hipMemset(buf, 0, sizeof(float) * 2 * nb * npb));
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      float ms_chain = 0, ms_launch = 0;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(counter, 0, 4)); CK(hipMemset(fail, 0, 4));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_chain, dim3(nb), dim3(512), 0, 0, buf, npb, counter, rounds, fail);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_chain, e0, e1));
        CK(hipEventRecord(e0));
        for (int r = 0; r < rounds; ++r) hipLaunchKernelGGL(k_round, dim3(nb), dim3(512), 0, 0, buf, npb, r);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_launch, e0, e1));
      }
      // the same chain of dependent kernels replayed from a hipGraph
      float ms_graph = 0;
      {
        hipStream_t cs; CK(hipStreamCreate(&cs));
        hipGraph_t graph; hipGraphExec_t exec;
        CK(hipStreamBeginCapture(cs, hipStreamCaptureModeGlobal));
        for (int r = 0; r < rounds; ++r) hipLaunchKernelGGL(k_round, dim3(nb), dim3(512), 0, cs, buf, npb, r);
        CK(hipStreamEndCapture(cs, &graph));
        CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        CK(hipGraphLaunch(exec, cs)); CK(hipStreamSynchronize(cs));
        CK(hipEventRecord(e0, cs));
        CK(hipGraphLaunch(exec, cs));
        CK(hipEventRecord(e1, cs)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_graph, e0, e1));
        CK(hipGraphExecDestroy(exec)); CK(hipGraphDestroy(graph)); CK(hipStreamDestroy(cs));
      }
      printf("   hipGraph replay of the same chain: %.2f us/round\n", ms_graph * 1e3 / rounds);
      int hfail = 0; float h0 = 0;
      CK(hipMemcpy(&hfail, fail, 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(&h0, buf, 4, hipMemcpyDeviceToHost));
      printf("blocks %3d  floats/block %5d : grid barrier %.2f us/round, kernel boundary %.2f us/round (fail=%d, value %.0f)\n",
             nb, npb, ms_chain * 1e3 / rounds, ms_launch * 1e3 / rounds, hfail, h0);
      CK(hipFree(buf)); CK(hipFree(counter)); CK(hipFree(fail));
    }
  }
  return 0;
}
