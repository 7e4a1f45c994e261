// oak_amd/csrc/oakgpu_internal.h -- shared between the translation units of liboakgpu.so.
#pragma once
#include <stddef.h>
struct oakgpu_ctx;
int oakgpu_fail_hip(int hip_error, const char *what); // records hipGetErrorString, returns the code
int oakgpu_fail_msg(const char *what);                 // records the message, returns -1
int oakgpu_ctx_device(const oakgpu_ctx *ctx);
void *oakgpu_ctx_stream(const oakgpu_ctx *ctx);        // hipStream_t
int oakgpu_ctx_enter(oakgpu_ctx *ctx);                 // hipSetDevice(ctx->device): first line of every entry point that launches or allocates
// Per-context device workspaces (slot 0: battle embeddings, 1: policy activations, 2: party-slot work list): grow-only, one per context = one
// per stream, so two contexts evaluating the same network never share scratch memory.  nullptr on failure (error recorded).
void *oakgpu_ctx_workspace(oakgpu_ctx *ctx, int slot, size_t bytes);
// Staging buffers of the host-pointer entry points: a grow-only cache owned by the context (slot k of a call = the k-th
// buffer it asks for).  A HostCall brackets one host-pointer call: its destructor synchronises the context's stream on
// EVERY exit path, so no async copy to / from the caller's buffers is still in flight when the call returns.
void *oakgpu_stage_get(oakgpu_ctx *ctx, size_t bytes); // nullptr on failure (error recorded)
void oakgpu_stage_begin(oakgpu_ctx *ctx);
void oakgpu_stage_end(oakgpu_ctx *ctx);
struct OakHostCall {
  oakgpu_ctx *c;
  explicit OakHostCall(oakgpu_ctx *ctx) : c(ctx) { oakgpu_stage_begin(c); }
  ~OakHostCall() { oakgpu_stage_end(c); }
  void *get(size_t bytes) { return oakgpu_stage_get(c, bytes); }
};
extern "C" int oakgpu_leaf_set_lds_limits(void);                // leafnet.hip: per-device kernel attributes (called by oakgpu_create)
// Optional per-kernel timing of the leaf evaluator (oakgpu_set_kernel_timing): 4 events = before the party-slot
// embedding pass, before the actives' pass, before the main net, after it.  nullptr when timing is off.
void **oakgpu_ctx_timing_events(oakgpu_ctx *ctx);
// One opaque attachment per context, freed (through its destructor) by oakgpu_destroy before the context's own resources:
// the tree search keeps its batch slots (second context, device arrays, pinned mirrors) here between searches.
void *oakgpu_ctx_attachment(const oakgpu_ctx *ctx);
void oakgpu_ctx_set_attachment(oakgpu_ctx *ctx, void *p, void (*dtor)(void *));
// A caller that keeps several contexts busy at the same time (the tree search: two batches in flight) says so: launches that
// do not fill the device then run in regrouping rounds, whose dispatch boundaries let the other context's small kernels in.
// Returns the previous value.
int oakgpu_ctx_set_concurrent_hint(oakgpu_ctx *ctx, int on);
// Host threads of the tree walks started by the CALLING thread (0 = the default rule): callers that run several searches side
// by side -- oakgpu_search_many, oakgpu_selfplay_games -- give each its share of the cores.
void oakgpu_set_thread_search_threads(int threads);
// Cores the process may really use (affinity mask capped by the cgroup CPU quota; OAKGPU_SEARCH_CORES overrides): search_host.hip.
unsigned oakgpu_usable_cores();
