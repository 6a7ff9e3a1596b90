// comm.hip — the cross-rank exchange step of a slab-sharded run.
//
// The reference has no counterpart (single process, OpenMP over <= 6 candidate poses: SURVEY.md 2); this is the
// build's multi-GPU row: every rank holds a slab of each source cloud, the target is replicated, and once per
// correspondence pass the ranks exchange a few doubles per job.
//
// Floating-point sums: the per-rank rows are ALL-GATHERED and every rank adds them in rank order 0, 1, 2, ...
// (rank_order_sum_kernel / the host loop below).  An all-reduce would leave the association order to the transport
// (ring vs tree, channel count), so 2-, 4- and 8-GPU runs of one workload could differ in the last bits and ranks
// of one run could in principle disagree; with the gather every rank computes the same expression.  The message is
// 136 B per job and rank - pure latency either way.
// Integer reductions (radix-select histograms, picky owner keys) are exact in any order: ncclAllReduce natively.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>

#include "icp_internal.h"

namespace mi355 {

namespace {

__global__ void rank_order_sum_kernel(const double* __restrict__ gathered, int n_ranks, uint32_t count, double* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  double acc = gathered[i];  // rank 0 first: with one rank the result is the input, bit for bit
  for (int r = 1; r < n_ranks; ++r) acc += gathered[static_cast<size_t>(r) * count + i];
  out[i] = acc;
}

// Up to 8 job descriptors travel as KERNEL ARGUMENTS (the packet's kernarg segment) and one wave writes them to the device
// tables: one launch instead of a host-to-device copy per table per pass - a copy is a blit kernel plus its ordering on the
// stream, several microseconds each, which is what a pass on a coarse pyramid level costs in total.
__global__ void stage_jobs_kernel(StageArgs a, JobDesc* __restrict__ jobs, RefJobDesc* __restrict__ rjobs) {
  constexpr uint32_t kJ = sizeof(JobDesc) / 4, kR = sizeof(RefJobDesc) / 4;
  const uint32_t* src = reinterpret_cast<const uint32_t*>(a.jobs);
  uint32_t* dst = reinterpret_cast<uint32_t*>(jobs + a.first);
  for (uint32_t w = threadIdx.x; w < a.n * kJ; w += blockDim.x) dst[w] = src[w];
  if (a.has_ref) {
    const uint32_t* rs = reinterpret_cast<const uint32_t*>(a.rjobs);
    uint32_t* rd = reinterpret_cast<uint32_t*>(rjobs + a.first);
    for (uint32_t w = threadIdx.x; w < a.n * kR; w += blockDim.x) rd[w] = rs[w];
  }
}

int hip_fail(hipError_t e, const char** err) {
  if (e == hipSuccess) return MI355_ICP_OK;
  if (err) *err = hipGetErrorString(e);
  return MI355_ICP_ERR_HIP;
}

int ensure_dev(CommState& cs, size_t bytes, const char** err) {
  if (bytes <= cs.gather_dev_cap) return MI355_ICP_OK;
  if (cs.gather_dev) { (void)hipFree(cs.gather_dev); cs.gather_dev = nullptr; cs.gather_dev_cap = 0; }
  const int st = hip_fail(hipMalloc(reinterpret_cast<void**>(&cs.gather_dev), bytes * 2), err);
  if (st == MI355_ICP_OK) cs.gather_dev_cap = bytes * 2;
  return st;
}

int ensure_stage(CommState& cs, size_t send_bytes, size_t recv_bytes, const char** err) {
  if (send_bytes > cs.stage_send_cap) {
    if (cs.stage_send) { (void)hipHostFree(cs.stage_send); cs.stage_send = nullptr; cs.stage_send_cap = 0; }
    const int st = hip_fail(hipHostMalloc(reinterpret_cast<void**>(&cs.stage_send), send_bytes * 2, hipHostMallocDefault), err);
    if (st != MI355_ICP_OK) return st;
    cs.stage_send_cap = send_bytes * 2;
  }
  if (recv_bytes > cs.stage_recv_cap) {
    if (cs.stage_recv) { (void)hipHostFree(cs.stage_recv); cs.stage_recv = nullptr; cs.stage_recv_cap = 0; }
    const int st = hip_fail(hipHostMalloc(reinterpret_cast<void**>(&cs.stage_recv), recv_bytes * 2, hipHostMallocDefault), err);
    if (st != MI355_ICP_OK) return st;
    cs.stage_recv_cap = recv_bytes * 2;
  }
  return MI355_ICP_OK;
}

// external transport: device -> pinned host, caller's all-gather, reduce on the host in rank order, host -> device
template <typename T, typename Op>
int external_reduce(CommState& cs, T* dev_buf, size_t count, void* stream, const char** err, Op op) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  const size_t bytes = count * sizeof(T);
  int st = ensure_stage(cs, bytes, bytes * static_cast<size_t>(cs.size), err);
  if (st != MI355_ICP_OK) return st;
  if ((st = hip_fail(hipMemcpyAsync(cs.stage_send, dev_buf, bytes, hipMemcpyDeviceToHost, s), err)) != MI355_ICP_OK) return st;
  if ((st = hip_fail(hipStreamSynchronize(s), err)) != MI355_ICP_OK) return st;
  if (cs.ext(cs.ext_user, cs.stage_send, cs.stage_recv, bytes) != 0) {
    if (err) *err = "the caller's all-gather reported a failure";
    return MI355_ICP_ERR_COMM;
  }
  T* out = reinterpret_cast<T*>(cs.stage_send);
  const T* in = reinterpret_cast<const T*>(cs.stage_recv);
  for (size_t i = 0; i < count; ++i) {
    T acc = in[i];
    for (int r = 1; r < cs.size; ++r) acc = op(acc, in[static_cast<size_t>(r) * count + i]);
    out[i] = acc;
  }
  return hip_fail(hipMemcpyAsync(dev_buf, cs.stage_send, bytes, hipMemcpyHostToDevice, s), err);
}

}  // namespace

int comm_sum_f64_rank_order(CommState& cs, double* dev_buf, size_t count, void* stream, const char** err) {
  if (!cs.attached() || count == 0) return MI355_ICP_OK;
  if (cs.ext) return external_reduce(cs, dev_buf, count, stream, err, [](double a, double b) { return a + b; });
  const size_t bytes = count * sizeof(double);
  int st = ensure_dev(cs, bytes * static_cast<size_t>(cs.size), err);
  if (st != MI355_ICP_OK) return st;
  if ((st = rccl_allgather_bytes(cs.rccl, dev_buf, cs.gather_dev, bytes, stream, err)) != MI355_ICP_OK) return st;
  const uint32_t n = static_cast<uint32_t>(count);
  hipLaunchKernelGGL(rank_order_sum_kernel, dim3((n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream),
                     reinterpret_cast<const double*>(cs.gather_dev), cs.size, n, dev_buf);
  return hip_fail(hipGetLastError(), err);
}

// The same sum for rows that are ALREADY on the host (a pass that ended itself wrote them to pinned memory: PassTail) - external
// transports only: one all-gather through the caller's callback, no device copy in either direction.  A rank whose pass did not end
// itself makes the device-side call above instead; both are one all-gather of count doubles per rank, so they match.
int comm_sum_f64_rank_order_host(CommState& cs, double* host_buf, size_t count, const char** err) {
  if (!cs.ext || count == 0) return MI355_ICP_OK;
  const size_t bytes = count * sizeof(double);
  const int st = ensure_stage(cs, bytes, bytes * static_cast<size_t>(cs.size), err);
  if (st != MI355_ICP_OK) return st;
  std::memcpy(cs.stage_send, host_buf, bytes);
  if (cs.ext(cs.ext_user, cs.stage_send, cs.stage_recv, bytes) != 0) {
    if (err) *err = "the caller's all-gather reported a failure";
    return MI355_ICP_ERR_COMM;
  }
  const double* in = reinterpret_cast<const double*>(cs.stage_recv);
  for (size_t i = 0; i < count; ++i) {
    double acc = in[i];
    for (int r = 1; r < cs.size; ++r) acc += in[static_cast<size_t>(r) * count + i];
    host_buf[i] = acc;
  }
  return MI355_ICP_OK;
}

int comm_sum_u32(CommState& cs, uint32_t* dev_buf, size_t count, void* stream, const char** err) {
  if (!cs.attached() || count == 0) return MI355_ICP_OK;
  if (cs.ext) return external_reduce(cs, dev_buf, count, stream, err, [](uint32_t a, uint32_t b) { return a + b; });
  return rccl_allreduce_u32_sum(cs.rccl, dev_buf, count, stream, err);
}

int comm_min_u64(CommState& cs, unsigned long long* dev_buf, size_t count, void* stream, const char** err) {
  if (!cs.attached() || count == 0) return MI355_ICP_OK;
  if (cs.ext) return external_reduce(cs, dev_buf, count, stream, err, [](unsigned long long a, unsigned long long b) { return a < b ? a : b; });
  return rccl_allreduce_u64_min(cs.rccl, dev_buf, count, stream, err);
}

int comm_allgather_device(CommState& cs, const void* dev_send, void* dev_recv, size_t bytes_per_rank, void* stream, const char** err) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (!cs.attached()) return hip_fail(hipMemcpyAsync(dev_recv, dev_send, bytes_per_rank, hipMemcpyDeviceToDevice, s), err);
  if (!cs.ext) return rccl_allgather_bytes(cs.rccl, dev_send, dev_recv, bytes_per_rank, stream, err);
  const size_t total = bytes_per_rank * static_cast<size_t>(cs.size);
  int st = ensure_stage(cs, bytes_per_rank, total, err);
  if (st != MI355_ICP_OK) return st;
  if ((st = hip_fail(hipMemcpyAsync(cs.stage_send, dev_send, bytes_per_rank, hipMemcpyDeviceToHost, s), err)) != MI355_ICP_OK) return st;
  if ((st = hip_fail(hipStreamSynchronize(s), err)) != MI355_ICP_OK) return st;
  if (cs.ext(cs.ext_user, cs.stage_send, cs.stage_recv, bytes_per_rank) != 0) {
    if (err) *err = "the caller's all-gather reported a failure";
    return MI355_ICP_ERR_COMM;
  }
  return hip_fail(hipMemcpyAsync(dev_recv, cs.stage_recv, total, hipMemcpyHostToDevice, s), err);
}

int comm_allgather_host(CommState& cs, const void* send, void* recv, size_t bytes_per_rank, void* stream, const char** err) {
  if (!cs.attached()) {
    std::memcpy(recv, send, bytes_per_rank);
    return MI355_ICP_OK;
  }
  if (cs.ext) {
    if (cs.ext(cs.ext_user, send, recv, bytes_per_rank) != 0) {
      if (err) *err = "the caller's all-gather reported a failure";
      return MI355_ICP_ERR_COMM;
    }
    return MI355_ICP_OK;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  const size_t total = bytes_per_rank * static_cast<size_t>(cs.size);
  int st = ensure_dev(cs, total + bytes_per_rank, err);
  if (st != MI355_ICP_OK) return st;
  unsigned char* dsend = cs.gather_dev + total;
  if ((st = hip_fail(hipMemcpyAsync(dsend, send, bytes_per_rank, hipMemcpyHostToDevice, s), err)) != MI355_ICP_OK) return st;
  if ((st = hip_fail(hipStreamSynchronize(s), err)) != MI355_ICP_OK) return st;  // `send` may be pageable
  if ((st = rccl_allgather_bytes(cs.rccl, dsend, cs.gather_dev, bytes_per_rank, stream, err)) != MI355_ICP_OK) return st;
  if ((st = hip_fail(hipMemcpyAsync(recv, cs.gather_dev, total, hipMemcpyDeviceToHost, s), err)) != MI355_ICP_OK) return st;
  return hip_fail(hipStreamSynchronize(s), err);
}

void comm_release(CommState& cs) {
  if (cs.rccl) rccl_comm_destroy(cs.rccl);
  (void)hipFree(cs.gather_dev);
  if (cs.stage_send) (void)hipHostFree(cs.stage_send);
  if (cs.stage_recv) (void)hipHostFree(cs.stage_recv);
  cs = CommState{};
}

// the same descriptors as one StageArgs value, for a launch that stages them itself (launch_nn_scan); false: too many
bool fill_stage_args(const JobDesc* jobs_host, const RefJobDesc* rjobs_host, int n, StageArgs* out) {
  if (n < 1 || n > 8) return false;
  *out = StageArgs{};
  out->n = static_cast<uint32_t>(n);
  out->has_ref = rjobs_host ? 1u : 0u;
  std::memcpy(out->jobs, jobs_host, static_cast<size_t>(n) * sizeof(JobDesc));
  if (rjobs_host) std::memcpy(out->rjobs, rjobs_host, static_cast<size_t>(n) * sizeof(RefJobDesc));
  return true;
}

// one system-scope store behind everything that is on the stream: what sync_pass (icp_engine.hip) waits for
__global__ void publish_word_kernel(uint32_t* word, uint32_t value) {
  __hip_atomic_store(word, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
void launch_publish_word(void* stream, uint32_t* mapped_word, uint32_t value) {
  hipLaunchKernelGGL(publish_word_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream), mapped_word, value);
}

int stage_jobs(void* stream, const JobDesc* jobs_host, const RefJobDesc* rjobs_host, int n, JobDesc* jobs_dev, RefJobDesc* rjobs_dev) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (n > 32) {  // large batches: plain copies from the pinned tables
    hipError_t e = hipMemcpyAsync(jobs_dev, jobs_host, static_cast<size_t>(n) * sizeof(JobDesc), hipMemcpyHostToDevice, s);
    if (e == hipSuccess && rjobs_host) e = hipMemcpyAsync(rjobs_dev, rjobs_host, static_cast<size_t>(n) * sizeof(RefJobDesc), hipMemcpyHostToDevice, s);
    return e == hipSuccess ? MI355_ICP_OK : MI355_ICP_ERR_HIP;
  }
  for (int first = 0; first < n; first += 8) {
    StageArgs a{};
    a.n = static_cast<uint32_t>(std::min(8, n - first));
    a.has_ref = rjobs_host ? 1u : 0u;
    a.first = static_cast<uint32_t>(first);
    std::memcpy(a.jobs, jobs_host + first, a.n * sizeof(JobDesc));
    if (rjobs_host) std::memcpy(a.rjobs, rjobs_host + first, a.n * sizeof(RefJobDesc));
    hipLaunchKernelGGL(stage_jobs_kernel, dim3(1), dim3(256), 0, s, a, jobs_dev, rjobs_dev);
  }
  return hipGetLastError() == hipSuccess ? MI355_ICP_OK : MI355_ICP_ERR_HIP;
}

}  // namespace mi355
