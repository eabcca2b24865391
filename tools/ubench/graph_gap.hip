// Micro-benchmark (measurement aid): what does the boundary BETWEEN two replays of a captured graph cost on one stream, next to
// the 1.7 us of a dependent launch INSIDE a graph (launch_chain.hip)?  The batch-32 DQN step is one update graph per step on the
// update stream (6 kernels, ~89 us of spans) beside one actor graph per step on a CU-masked second stream that waits for the
// previous update's event; the rocprofv3 timeline shows ~16 us between the last kernel of update t and the first of update t+1.
// This program rebuilds that shape from timed spin kernels and varies one thing at a time:
//   same_stream        graph of 6 kernels replayed back to back, nothing in between
//   +record            ... an event record after every replay (what the learner does: ev_upd[q])
//   +actor             ... a second CU-masked stream: wait(previous update's event), one 78 us kernel graph, event record
//   +hostsync          ... the host never runs more than 3 steps ahead (hipEventSynchronize on the event of step t-4)
//   eager              the 6 kernels as plain launches, no graph
//   two_per_graph      two updates captured as ONE graph (12 kernels), the actor waits only every other step
// Reported: microseconds per step minus the sum of the kernels' programmed spans = boundary cost per step.
// build: hipcc --offload-arch=gfx950 -O3 -o graph_gap tools/ubench/graph_gap.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// every workgroup spins until `ticks` of the 100 MHz clock have passed since ITS start
__global__ void __launch_bounds__(256) spin_kernel(int ticks, float* sink) {
  const unsigned long long t0 = __builtin_readcyclecounter();
  (void)t0;
  const unsigned long long s = wall_clock64();
  while (wall_clock64() - s < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(2);
  if (sink == nullptr) __builtin_trap();
}

// the update's LAST kernel in the flag forms: the last workgroup to finish publishes the step number to device memory (the actor
// kernel polls it) and to pinned host memory (the host polls it): no event record, no stream wait, no event synchronize
struct Pub { unsigned* arrive; unsigned long long* seq_dev; volatile unsigned long long* seq_host; unsigned long long* counter; };
__global__ void __launch_bounds__(256) spin_pub_kernel(int ticks, Pub p) {
  const unsigned long long s = wall_clock64();
  while (wall_clock64() - s < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(2);
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned n = __hip_atomic_fetch_add(p.arrive, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (n == gridDim.x - 1) {
      __hip_atomic_store(p.arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned long long v = *p.counter + 1;
      *p.counter = v;
      __hip_atomic_store(p.seq_dev, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store((unsigned long long*)p.seq_host, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}
// the actor in the flag forms: launch number n (its own device counter) waits until the update has published n - lag
__global__ void __launch_bounds__(512) spin_wait_kernel(int ticks, const unsigned long long* seq_dev, unsigned long long* my_count, int lag, int* fail) {
  __shared__ unsigned long long s_n;
  if (threadIdx.x == 0) {
    const unsigned long long n = *my_count;      // (bumped by workgroup 0 at the END of the previous launch)
    s_n = n;
    long spins = 0;
    while ((long long)__hip_atomic_load(seq_dev, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (long long)n - lag) {
      __builtin_amdgcn_s_sleep(4);
      if (++spins > 50000000) { *fail = 1; break; }
    }
  }
  __syncthreads();
  const unsigned long long s = wall_clock64();
  while (wall_clock64() - s < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(2);
  __syncthreads();
  if (blockIdx.x == 0 && threadIdx.x == 0) *my_count = s_n + 1;   // (every workgroup of THIS launch has read it: they all started before ticks elapsed)
}

static const int kSpans[6] = {3200, 800, 700, 850, 2600, 800};   // ticks of 10 ns: the update graph's six kernels
static const int kWgs[6] = {1995, 224, 130, 314, 1458, 153};
static const int kActor = 7800;

struct Ctx {
  hipStream_t su, sa;
  float* sink;
  hipGraphExec_t gu, gu2, ga, gu_pub, ga_wait, gu_one, ga_free;
  Pub pub; unsigned long long* a_count; int* fail; volatile unsigned long long* seq_host;
  std::vector<hipEvent_t> eu, ea;
};

static void launch_update(Ctx& c, hipStream_t st) {
  for (int k = 0; k < 6; ++k) hipLaunchKernelGGL(spin_kernel, dim3(kWgs[k] > 448 ? 448 : kWgs[k]), dim3(256), 0, st, kSpans[k], c.sink);
}

static void launch_update_pub(Ctx& c, hipStream_t st) {
  for (int k = 0; k < 5; ++k) hipLaunchKernelGGL(spin_kernel, dim3(kWgs[k] > 448 ? 448 : kWgs[k]), dim3(256), 0, st, kSpans[k], c.sink);
  hipLaunchKernelGGL(spin_pub_kernel, dim3(kWgs[5]), dim3(256), 0, st, kSpans[5], c.pub);
}

// kind 0: n updates, 1: actor, 2: update with publishing tail, 3: actor that polls the published step, 4: ONE kernel as long as the six
static hipGraphExec_t capture(Ctx& c, hipStream_t st, int n_updates, int kind) {
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  if (kind == 1) hipLaunchKernelGGL(spin_kernel, dim3(32), dim3(512), 0, st, kActor, c.sink);
  else if (kind == 2) launch_update_pub(c, st);
  else if (kind == 3) hipLaunchKernelGGL(spin_wait_kernel, dim3(32), dim3(512), 0, st, kActor, (const unsigned long long*)c.pub.seq_dev, c.a_count, 0, c.fail);
  else if (kind == 4) hipLaunchKernelGGL(spin_kernel, dim3(448), dim3(256), 0, st, 8950, c.sink);
  else for (int i = 0; i < n_updates; ++i) launch_update(c, st);
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  CK(hipGraphDestroy(g));
  return ge;
}

// mode bits: 1 record, 2 actor stream, 4 host sync (3 ahead), 8 eager, 16 two per graph, 32 actor never waits, 64 flags instead
// of events (the host polls pinned memory, the actor kernel polls device memory), 128 one kernel instead of six
static double run(Ctx& c, int mode, int steps) {
  const bool rec = mode & 1, actor = mode & 2, hsync = mode & 4, eager = mode & 8, two = mode & 16, nowait = mode & 32, flags = mode & 64, one = mode & 128;
  if (flags) {
    CK(hipDeviceSynchronize());
    CK(hipMemset(c.pub.seq_dev, 0, 8)); CK(hipMemset(c.pub.counter, 0, 8)); CK(hipMemset(c.a_count, 0, 8)); *c.seq_host = 0;
    CK(hipDeviceSynchronize());
    unsigned long long issued = 0;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int pass = 0; pass < 2; ++pass) {
      const int n = pass == 0 ? 40 : steps;
      if (pass == 1) { CK(hipDeviceSynchronize()); CK(hipEventRecord(e0, c.su)); }
      for (int t = 0; t < n; ++t) {
        if (hsync) while (*c.seq_host + 3 < issued) __builtin_ia32_pause();
        if (eager) launch_update_pub(c, c.su); else CK(hipGraphLaunch(c.gu_pub, c.su));
        ++issued;
        if (actor) {
          if (eager) hipLaunchKernelGGL(spin_wait_kernel, dim3(32), dim3(512), 0, c.sa, kActor, (const unsigned long long*)c.pub.seq_dev, c.a_count, 0, c.fail);
          else CK(hipGraphLaunch(c.ga_wait, c.sa));
        }
      }
    }
    CK(hipEventRecord(e1, c.su));
    CK(hipEventSynchronize(e1));
    CK(hipDeviceSynchronize());
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return (double)ms * 1e3 / steps;
  }
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int pass = 0; pass < 2; ++pass) {
    const int n = pass == 0 ? 40 : steps;
    if (pass == 1) { CK(hipDeviceSynchronize()); CK(hipEventRecord(e0, c.su)); }
    for (int t = 0; t < n; ++t) {
      const int q = t & 7;
      if (hsync && t >= 4) CK(hipEventSynchronize(c.eu[(t - 4) & 7]));
      if (eager) launch_update(c, c.su);
      else if (two) { if (!(t & 1)) CK(hipGraphLaunch(c.gu2, c.su)); }
      else CK(hipGraphLaunch(one ? c.gu_one : c.gu, c.su));
      if (rec || actor || hsync) CK(hipEventRecord(c.eu[q], c.su));
      if (actor && (!two || !(t & 1))) {
        if (t > 0 && !nowait) CK(hipStreamWaitEvent(c.sa, c.eu[(t - 1) & 7], 0));
        CK(hipGraphLaunch(c.ga, c.sa));
        if (two) CK(hipGraphLaunch(c.ga, c.sa));
        CK(hipEventRecord(c.ea[q], c.sa));
      }
    }
  }
  CK(hipEventRecord(e1, c.su));
  CK(hipEventSynchronize(e1));
  CK(hipDeviceSynchronize());
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
  return (double)ms * 1e3 / steps;
}

int main(int argc, char** argv) {
  const int steps = argc > 1 ? atoi(argv[1]) : 400;
  Ctx c;
  uint32_t mask_u[8], mask_a[8];
  for (int w = 0; w < 8; ++w) { mask_u[w] = 0xffffffffu; mask_a[w] = 0; }
  // actor partition: 4 CUs of each XCD (CU index i lives on XCD i % 8): bits 0..31; the update stream gets the rest
  mask_a[0] = 0xffffffffu; mask_u[0] = 0;
  CK(hipExtStreamCreateWithCUMask(&c.su, 8, mask_u));
  CK(hipExtStreamCreateWithCUMask(&c.sa, 8, mask_a));
  CK(hipMalloc(&c.sink, 256));
  c.eu.resize(8); c.ea.resize(8);
  for (int i = 0; i < 8; ++i) { CK(hipEventCreateWithFlags(&c.eu[i], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&c.ea[i], hipEventDisableTiming)); }
  CK(hipMalloc(&c.pub.arrive, 256)); CK(hipMemset(c.pub.arrive, 0, 256));
  CK(hipMalloc(&c.pub.seq_dev, 256)); CK(hipMalloc(&c.pub.counter, 256)); CK(hipMalloc(&c.a_count, 256));
  CK(hipHostMalloc((void**)&c.seq_host, 256, hipHostMallocCoherent)); c.pub.seq_host = c.seq_host;
  CK(hipHostMalloc((void**)&c.fail, 256, hipHostMallocCoherent)); *c.fail = 0;
  c.gu = capture(c, c.su, 1, 0);
  c.gu2 = capture(c, c.su, 2, 0);
  c.ga = capture(c, c.sa, 1, 1);
  c.gu_pub = capture(c, c.su, 1, 2);
  c.ga_wait = capture(c, c.sa, 1, 3);
  c.gu_one = capture(c, c.su, 1, 4);
  double sum = 0; for (int k = 0; k < 6; ++k) sum += kSpans[k] * 0.01;
  struct { const char* name; int mode; } cases[] = {
    {"same_stream", 0}, {"record", 1}, {"record_actor", 3}, {"record_actor_hostsync", 7}, {"record_hostsync", 5},
    {"eager", 8}, {"eager_record_actor_hostsync", 15}, {"two_per_graph", 16}, {"two_per_graph_actor_hostsync", 23},
    {"record_actor_nowait", 35}, {"one_kernel_graph", 128}, {"one_kernel_graph_record", 129},
    {"flags", 64}, {"flags_hostsync", 68}, {"flags_actor_hostsync", 70}, {"flags_actor_hostsync_eager", 78}};
  printf("{\"steps\": %d, \"kernel_spans_us\": %.1f, \"actor_span_us\": %.1f", steps, sum, kActor * 0.01);
  for (int rep = 0; rep < 2; ++rep)
    for (auto& cs : cases) {
      const double us = run(c, cs.mode, steps);
      printf(", \"%s_%d\": {\"us_per_step\": %.2f, \"boundary_us\": %.2f}", cs.name, rep, us, us - sum);
      fflush(stdout);
    }
  printf(", \"fail\": %d}\n", *c.fail);
  return 0;
}
