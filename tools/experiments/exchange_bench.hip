// tools/experiments/exchange_bench.hip -- what would it cost to move every playout's state through memory once per turn?
//
// DESIGN 3 / VERDICT r2 #3 ask for "waves that do one thing": device-wide per-class queues from which any wave pulls 64 records
// of ONE action class, the state travelling through memory between turns instead of staying in the lane's registers.  Before
// building that engine this micro-benchmark prices its skeleton on the real machine: a persistent grid of 4 waves per SIMD;
// per iteration ("turn") a wave (1) reserves 64 slots of a class ring with one atomic, (2) stores its lanes' records -- NW
// dwords per lane, SoA so that every store instruction writes 1 KB contiguous -- write-through (sc1), drains them and publishes
// a per-slot sequence word, (3) reserves 64 slots to consume with another atomic and loads those records (sc1: L1 bypassed),
// (4) executes WORK dependent VALU instructions (the turn).  Slots to consume are ones written two iterations earlier by
// another wave, so no wave ever waits: the numbers are the cost of the TRAFFIC and the ATOMICS, without the queueing delays a
// real engine adds on top.  mode 0: no exchange (the register-resident engine's shape); 1: plain loads / stores; 2: sc1.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/exchange_bench tools/experiments/exchange_bench.hip && /tmp/exchange_bench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int NW4, int MODE>
__global__ __launch_bounds__(64, 4) void k_xch(u32x4 *ring, uint32_t *seq, uint32_t *counters, uint32_t slots, int iters, int work, uint32_t *sink, int local) {
  const uint32_t lane = threadIdx.x, w = blockIdx.x, waves = gridDim.x;
  // local: every group of waves that shares an XCD (blockIdx % 8, MI355X_MICROARCH.md: dispatch is round-robin over the XCDs)
  // exchanges only inside its own region of the rings -- with plain stores the records then stay in that XCD's L2
  const uint32_t grp = local ? (w & 7u) : 0u, ngrp = local ? 8u : 1u;
  u32x4 st[NW4];
#pragma unroll
  for (int c = 0; c < NW4; ++c) st[c] = u32x4{w * 64 + lane + c, lane * 7 + c, w ^ c, 0x9E3779B9u * (c + 1)};
  uint32_t acc = lane;
  for (int it = 0; it < iters; ++it) {
    // ---- the "turn": WORK x 8 dependent integer instructions over the record ----
    uint32_t x = st[0].x ^ acc, y = st[NW4 - 1].w + it;
#pragma unroll 4
    for (int k = 0; k < work; ++k) {
      x ^= x << 13; y += x; x ^= x >> 17; y ^= y << 5; x += y; x ^= x << 5; y ^= x >> 3; y += 0x85EBCA6Bu;
    }
    st[0].x = x; st[NW4 - 1].w = y;
    acc += x ^ y;
    if (MODE == 0) continue;
    const uint32_t cls = (w + (uint32_t)it) % 9u;
    // ---- push: one atomic per wave, 1 KB per store instruction, drained, then the slots' sequence words ----
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(counters + (grp * 40 + cls) * 32, 64u);
    base = __shfl(base, 0, 64);
    const uint32_t slot = (grp * 9 + cls) * slots + ((base + lane) & (slots - 1));
#pragma unroll
    for (int c = 0; c < NW4; ++c) {
      u32x4 *p = ring + (size_t)c * 9 * ngrp * slots + slot;
      if (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(st[c]) : "memory");
      else *p = st[c];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (MODE == 2) __hip_atomic_store(seq + slot, base + lane + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else seq[slot] = base + lane + 1;
    // ---- pop: one atomic per wave; the records another wave pushed two iterations ago ----
    const uint32_t cls2 = (w + 5u + (uint32_t)it) % 9u;
    uint32_t base2 = 0;
    if (lane == 0) base2 = atomicAdd(counters + (grp * 40 + 16 + cls2) * 32, 64u);
    base2 = __shfl(base2, 0, 64);
    const uint32_t slot2 = (grp * 9 + cls2) * slots + ((base2 + lane + (slots >> 1)) & (slots - 1));
    uint32_t sq;
    if (MODE == 2) sq = __hip_atomic_load(seq + slot2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else sq = seq[slot2];
    acc += sq;
#pragma unroll
    for (int c = 0; c < NW4; ++c) {
      const u32x4 *p = ring + (size_t)c * 9 * ngrp * slots + slot2;
      if (MODE == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(st[c]) : "v"(p) : "memory");
      else st[c] = *p;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  uint32_t s = acc;
#pragma unroll
  for (int c = 0; c < NW4; ++c) s += st[c].x ^ st[c].y ^ st[c].z ^ st[c].w;
  if (s == 0x12345678u) sink[w * 64 + lane] = s; // (never true in practice: keeps the work alive)
  (void)waves;
}

template <int NW4, int MODE>
static double run(u32x4 *ring, uint32_t *seq, uint32_t *counters, uint32_t slots, int waves, int iters, int work, uint32_t *sink, int local) {
  CHECK(hipMemset(counters, 0, 8 * 40 * 32 * 4));
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  hipLaunchKernelGGL((k_xch<NW4, MODE>), dim3(waves), dim3(64), 0, 0, ring, seq, counters, slots, 8, work, sink, local); // warm
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(a));
  hipLaunchKernelGGL((k_xch<NW4, MODE>), dim3(waves), dim3(64), 0, 0, ring, seq, counters, slots, iters, work, sink, local);
  CHECK(hipEventRecord(b));
  CHECK(hipEventSynchronize(b));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, a, b));
  return ms;
}

int main(int argc, char **argv) {
  const int waves = 256 * 4 * 4, iters = argc > 1 ? atoi(argv[1]) : 400;
  // slots per class ring: 65536 (default; 9 x 64 Ki x 192 B = 113 MB: beyond every L2, inside the Infinity Cache), or per XCD region
  // with `local`: 4096 -> 7 MB per XCD (an engine needs slots >= the lanes in flight: 32,768 per XCD at 4 waves per SIMD =
  // 6.3 MB), 1024 -> 1.8 MB per XCD (fits the 4 MB L2; a real engine could not run with so few)
  const uint32_t slots = argc > 2 ? (uint32_t)atoi(argv[2]) : 65536;
  const int local = argc > 3 ? atoi(argv[3]) : 0;
  constexpr int NW4 = 12;       // 48 dwords per lane = the register engine's travelling state (two side frames + battle scalars + RNGs)
  u32x4 *ring; uint32_t *seq, *counters, *sink;
  const size_t ng = local ? 8 : 1;
  CHECK(hipMalloc(&ring, (size_t)NW4 * 9 * ng * slots * 16));
  CHECK(hipMalloc(&seq, (size_t)9 * ng * slots * 4));
  CHECK(hipMalloc(&counters, 8 * 40 * 32 * 4));
  CHECK(hipMalloc(&sink, (size_t)waves * 64 * 4));
  CHECK(hipMemset(ring, 0, (size_t)NW4 * 9 * ng * slots * 16));
  CHECK(hipMemset(seq, 0, (size_t)9 * ng * slots * 4));
  printf("%d waves (4 per SIMD), %d turns per wave, record %d dwords per lane, %u slots per class ring%s\n", waves, iters, NW4 * 4, slots, local ? ", one set of rings per XCD group" : "");
  printf("%-34s %8s %10s %14s %12s\n", "mode", "work", "ms", "G lane-turns/s", "us per turn");
  const int works[] = {560, 310, 200, 0}; // x 8 instructions: 4480 (today's wave-step), 2720, 2480, 2240, 1600, 800, 0
  for (int wk : works) {
    struct { const char *name; double ms; } r[3] = {
        {"0 no exchange (registers)", run<NW4, 0>(ring, seq, counters, slots, waves, iters, wk, sink, local)},
        {"1 exchange, plain loads/stores", run<NW4, 1>(ring, seq, counters, slots, waves, iters, wk, sink, local)},
        {"2 exchange, sc1 (coherent)", run<NW4, 2>(ring, seq, counters, slots, waves, iters, wk, sink, local)}};
    for (auto &x : r)
      printf("%-34s %8d %10.3f %14.2f %12.2f\n", x.name, wk * 8, x.ms, (double)waves * 64 * iters / x.ms / 1e6, x.ms * 1e3 / iters);
  }
  return 0;
}
