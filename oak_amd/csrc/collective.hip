// oak_amd/csrc/collective.hip -- the path's ONE exchange step, callable from the C++ host layer: per-root pre-reduction of
// the leaf values on the device and an RCCL all-gather of the result over xGMI.
//
// The reference has no collective (its playouts are unrelated std::threads, cpp/src/generate.cc:527-536); the analogue here
// is root-parallel MCTS sharded contiguous-by-root over the GPUs of a node: every rank rolls out the playouts of its own
// roots, reduces them to one mean value per root ON THE DEVICE (k_segment_mean: 4096 playouts -> 1 float), and ONE
// ncclAllGather of those means (256 floats for BASELINE config 4: latency-bound, not link-bound) gives every rank every
// root's value.  RCCL is bound at run time (dlopen of librccl.so, whichever copy the process already holds -- torch's
// when the caller is a torch program), so the library itself still loads on a machine without RCCL; every oakgpu_comm_*
// call then fails loudly.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <mutex>
#include <string>

#include "../../include/oakgpu.h"
#include "oakgpu_internal.h"

namespace oak {

// mean of each contiguous segment of `per` values; one workgroup per segment, coalesced float4 loads when aligned
__global__ __launch_bounds__(256) void k_segment_mean(const float *values, uint32_t segments, uint32_t per, float *out) {
  const uint32_t s = blockIdx.x;
  if (s >= segments) return;
  const float *v = values + (size_t)s * per;
  float acc = 0.0f;
  for (uint32_t i = threadIdx.x; i < per; i += 256) acc += v[i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  __shared__ float part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[s] = (part[0] + part[1] + part[2] + part[3]) / (float)per;
}

} // namespace oak

namespace {

// the few RCCL entry points used, resolved at run time
typedef struct { char internal[128]; } rccl_unique_id; // ncclUniqueId (rccl.h:43)
struct Rccl {
  void *lib = nullptr;
  int (*get_unique_id)(rccl_unique_id *) = nullptr;
  int (*comm_init_rank)(void **, int, rccl_unique_id, int) = nullptr;
  int (*comm_destroy)(void *) = nullptr;
  int (*all_gather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
  const char *(*error_string)(int) = nullptr;
  std::string why;
};
Rccl &rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    // the copy the process ALREADY holds first (RTLD_NOLOAD matches by SONAME: torch's bundled RCCL is "librccl.so.1"), so a
    // torch program never ends up with two RCCL runtimes; only then the search path, versioned name before the dev symlink
    r.lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
    if (!r.lib) r.lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      if (r.lib) break;
      r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    }
    if (!r.lib) { r.why = std::string("RCCL not found: ") + dlerror(); return; }
    r.get_unique_id = (int (*)(rccl_unique_id *))dlsym(r.lib, "ncclGetUniqueId");
    r.comm_init_rank = (int (*)(void **, int, rccl_unique_id, int))dlsym(r.lib, "ncclCommInitRank");
    r.comm_destroy = (int (*)(void *))dlsym(r.lib, "ncclCommDestroy");
    r.all_gather = (int (*)(const void *, void *, size_t, int, void *, hipStream_t))dlsym(r.lib, "ncclAllGather");
    r.error_string = (const char *(*)(int))dlsym(r.lib, "ncclGetErrorString");
    if (!r.get_unique_id || !r.comm_init_rank || !r.comm_destroy || !r.all_gather) { r.why = "RCCL library lacks an expected symbol"; r.lib = nullptr; }
  });
  return r;
}
int rccl_fail(int code, const char *what) {
  Rccl &r = rccl();
  return oakgpu_fail_msg((std::string(what) + ": " + (r.error_string ? r.error_string(code) : "RCCL error")).c_str());
}
constexpr int RCCL_FLOAT32 = 7; // ncclFloat32 (rccl.h ncclDataType_t)

} // namespace

struct oakgpu_comm {
  void *comm;
  int rank, world, device;
};

extern "C" {

int oakgpu_segment_mean_dev(oakgpu_ctx *ctx, const float *values, uint32_t segments, uint32_t per_segment, float *out) {
  if (!ctx) return oakgpu_fail_msg("null ctx");
  if (segments == 0) return 0;
  if (!values || !out || per_segment == 0) return oakgpu_fail_msg("oakgpu_segment_mean_dev: bad argument");
  if (int rc = oakgpu_ctx_enter(ctx)) return rc;
  hipLaunchKernelGGL(oak::k_segment_mean, dim3(segments), dim3(256), 0, (hipStream_t)oakgpu_ctx_stream(ctx), values, segments, per_segment, out);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : oakgpu_fail_hip((int)e, "k_segment_mean");
}

int oakgpu_comm_unique_id(uint8_t *id128) {
  if (!id128) return oakgpu_fail_msg("oakgpu_comm_unique_id: null argument");
  Rccl &r = rccl();
  if (!r.lib) return oakgpu_fail_msg(r.why.c_str());
  rccl_unique_id id;
  if (int rc = r.get_unique_id(&id)) return rccl_fail(rc, "ncclGetUniqueId");
  memcpy(id128, id.internal, 128);
  return 0;
}

int oakgpu_comm_create(oakgpu_ctx *ctx, const uint8_t *id128, int rank, int world, oakgpu_comm **out) {
  if (!ctx || !id128 || !out || world < 1 || rank < 0 || rank >= world) return oakgpu_fail_msg("oakgpu_comm_create: bad argument");
  Rccl &r = rccl();
  if (!r.lib) return oakgpu_fail_msg(r.why.c_str());
  if (int rc = oakgpu_ctx_enter(ctx)) return rc; // the communicator binds to the context's device
  rccl_unique_id id;
  memcpy(id.internal, id128, 128);
  void *comm = nullptr;
  if (int rc = r.comm_init_rank(&comm, world, id, rank)) return rccl_fail(rc, "ncclCommInitRank");
  *out = new oakgpu_comm{comm, rank, world, oakgpu_ctx_device(ctx)};
  return 0;
}

void oakgpu_comm_destroy(oakgpu_comm *c) {
  if (!c) return;
  Rccl &r = rccl();
  if (r.lib && c->comm) {
    // every collective issued on this communicator must have drained before it goes away, and its teardown (proxy
    // threads, registered buffers) must have finished before the caller destroys the streams / contexts it ran on
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    (void)r.comm_destroy(c->comm);
    (void)hipDeviceSynchronize();
  }
  delete c;
}

int oakgpu_all_gather_dev(oakgpu_ctx *ctx, oakgpu_comm *comm, const float *send, float *recv, size_t count) {
  if (!ctx || !comm || !send || !recv) return oakgpu_fail_msg("oakgpu_all_gather_dev: null argument");
  if (count == 0) return 0;
  Rccl &r = rccl();
  if (!r.lib) return oakgpu_fail_msg(r.why.c_str());
  if (int rc = oakgpu_ctx_enter(ctx)) return rc;
  if (int rc = r.all_gather(send, recv, count, RCCL_FLOAT32, comm->comm, (hipStream_t)oakgpu_ctx_stream(ctx))) return rccl_fail(rc, "ncclAllGather");
  return 0;
}

} // extern "C"
