// tools/gather_probe.hip -- what one 16-byte gather instruction costs a gfx950 CU, by the number and the pattern of
// its active lanes (is the price per instruction, per quad of lanes, or per cache line touched?).
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/gather_probe tools/gather_probe.hip
//   tools/bin/gather_probe [json-lines to stdout]
//
// Every wavefront runs ITERS rounds; a round issues INFLIGHT independent global_load_dwordx4 per active lane and the
// addresses of the next round depend on the data of this one (the shape of a tree descent: a chain of dependent
// gathers per lane, latency hidden only by the other wavefronts of the CU).  What varies:
//   lanes      which lanes of the wavefront are active (the others have left the kernel: exec mask)
//   share      lanes in units of 1 << share read consecutive 16-byte records of one random place (share 0: every lane
//              its own place; 2: a quad reads 64 contiguous bytes; 3: eight lanes one 128-byte line; 6: the wavefront
//              1 KB); `same`: all lanes one address
//   table      the bytes the random places are drawn from (L1 / L2 / Infinity Cache / HBM resident)
//   waves/CU   resident wavefronts (dynamic LDS as ballast)
// Reported: cycles of a CU per wave-level instruction = time x clock x CUs / (wavefronts x rounds x INFLIGHT).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#define CHECK(x)                                                                        \
  do {                                                                                  \
    hipError_t e_ = (x);                                                                \
    if (e_ != hipSuccess) {                                                             \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(1);                                                                          \
    }                                                                                   \
  } while (0)

extern __shared__ unsigned char ballast[];

template <int INFLIGHT>
__global__ __launch_bounds__(64) void gather_kernel(
    const uint4* __restrict__ tab, uint32_t n_mask, uint32_t iters, uint64_t lanes, uint32_t share, uint32_t same,
    uint32_t* __restrict__ sink) {
  const uint32_t lane = threadIdx.x;
  if (!((lanes >> lane) & 1ull)) return;
  const uint32_t unit = same ? 0u : lane >> share;
  const uint32_t low = same ? 0u : lane & ((1u << share) - 1u);
  uint32_t s[INFLIGHT];
#pragma unroll
  for (int u = 0; u < INFLIGHT; ++u) s[u] = (blockIdx.x * 64u + unit) * 0x9E3779B1u + (uint32_t)u * 0x85EBCA6Bu + 12345u;
  uint32_t acc = 0;
  for (uint32_t it = 0; it < iters; ++it) {
    uint4 v[INFLIGHT];
#pragma unroll
    for (int u = 0; u < INFLIGHT; ++u) {
      const uint32_t h = s[u] ^ (s[u] >> 15);
      const uint32_t idx = (((h >> 4) << share) | low) & n_mask;
      v[u] = tab[idx];
    }
#pragma unroll
    for (int u = 0; u < INFLIGHT; ++u) {
      acc += v[u].x + v[u].w + v[u].z;
      s[u] = s[u] * 1664525u + 1013904223u + v[u].y;  // (the table is zero: the chain is the generator's)
    }
  }
  if (acc == 0xDEADBEEFu) sink[0] = acc;
}

// The same rounds with a 12-byte and an 8-byte and a 4-byte load (is the price per byte of a lane?).
template <int WORDS>
__global__ __launch_bounds__(64) void gather_narrow_kernel(
    const uint32_t* __restrict__ tab, uint32_t n_mask, uint32_t iters, uint64_t lanes, uint32_t* __restrict__ sink) {
  const uint32_t lane = threadIdx.x;
  if (!((lanes >> lane) & 1ull)) return;
  uint32_t s = (blockIdx.x * 64u + lane) * 0x9E3779B1u + 12345u;
  uint32_t acc = 0;
  for (uint32_t it = 0; it < iters; ++it) {
    const uint32_t h = s ^ (s >> 15);
    const uint32_t idx = ((h >> 4) & n_mask) * 4u;
    uint32_t y = 0;
    if (WORDS == 1) {
      y = tab[idx];
    } else if (WORDS == 2) {
      const uint2 v = *reinterpret_cast<const uint2*>(tab + idx);
      y = v.x + v.y;
    } else {
      const uint3 v = *reinterpret_cast<const uint3*>(tab + idx);
      y = v.x + v.y + v.z;
    }
    acc += y;
    s = s * 1664525u + 1013904223u + y;
  }
  if (acc == 0xDEADBEEFu) sink[0] = acc;
}


// MODE 1: the INFLIGHT loads of a round read CONSECUTIVE 16-byte records of the lane's place (one line: what does a
// load cost that hits a line another load of the same wavefront has just missed on?).
// MODE 2: a round is a load at a random place and then -- dependent on it -- the neighbouring record of the same
// line (a tree descent whose child sits next to its parent: is the line still in the L1 one step later?).
template <int INFLIGHT, int MODE>
__global__ __launch_bounds__(64) void gather_mode_kernel(
    const uint4* __restrict__ tab, uint32_t n_mask, uint32_t iters, uint64_t lanes, uint32_t* __restrict__ sink) {
  const uint32_t lane = threadIdx.x;
  if (!((lanes >> lane) & 1ull)) return;
  uint32_t s = (blockIdx.x * 64u + lane) * 0x9E3779B1u + 12345u;
  uint32_t acc = 0;
  for (uint32_t it = 0; it < iters; ++it) {
    const uint32_t h = s ^ (s >> 15);
    if (MODE == 1) {
      const uint32_t base = ((h >> 4) * (uint32_t)INFLIGHT) & n_mask;
      uint4 v[INFLIGHT];
#pragma unroll
      for (int u = 0; u < INFLIGHT; ++u) v[u] = tab[base + u];
      uint32_t y = 0;
#pragma unroll
      for (int u = 0; u < INFLIGHT; ++u) {
        acc += v[u].x + v[u].w + v[u].z;
        y += v[u].y;
      }
      s = s * 1664525u + 1013904223u + y;
    } else {
      const uint32_t idx = (h >> 4) & n_mask;
      const uint4 a = tab[idx];
      const uint4 b = tab[(idx ^ 1u) + a.y];
      acc += a.x + a.w + a.z + b.x + b.w + b.z;
      s = s * 1664525u + 1013904223u + b.y;
    }
  }
  if (acc == 0xDEADBEEFu) sink[0] = acc;
}

struct Case {
  const char* name;
  uint64_t lanes;
  uint32_t share;
  uint32_t same;
};

static uint64_t every(int step, int n) {  // n lanes, one every `step`
  uint64_t m = 0;
  for (int i = 0; i < n; ++i) m |= 1ull << (i * step);
  return m;
}

int main(int argc, char** argv) {
  int dev = 0;
  CHECK(hipSetDevice(dev));
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, dev));
  const int cus = prop.multiProcessorCount;
  const double clock_hz = (double)prop.clockRate * 1e3;
  const size_t max_bytes = 1ull << 30;
  uint4* tab = nullptr;
  uint32_t* sink = nullptr;
  CHECK(hipMalloc(&tab, max_bytes));
  CHECK(hipMemset(tab, 0, max_bytes));
  CHECK(hipMalloc(&sink, 64));
  CHECK(hipMemset(sink, 0, 64));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));

  const Case cases[] = {
      {"64 lanes, own place", ~0ull, 0, 0},
      {"32 lanes (low half), own place", 0xFFFFFFFFull, 0, 0},
      {"32 lanes (every other), own place", every(2, 32), 0, 0},
      {"21 lanes (every third), own place", every(3, 21), 0, 0},
      {"16 lanes (low quarter), own place", 0xFFFFull, 0, 0},
      {"16 lanes (one per quad), own place", every(4, 16), 0, 0},
      {"8 lanes (low), own place", 0xFFull, 0, 0},
      {"8 lanes (every eighth), own place", every(8, 8), 0, 0},
      {"4 lanes (low), own place", 0xFull, 0, 0},
      {"1 lane", 1ull, 0, 0},
      {"64 lanes, pairs share 32 B", ~0ull, 1, 0},
      {"64 lanes, quads share 64 B", ~0ull, 2, 0},
      {"64 lanes, eights share a 128-B line", ~0ull, 3, 0},
      {"64 lanes, sixteens share 256 B", ~0ull, 4, 0},
      {"64 lanes, contiguous 1 KB", ~0ull, 6, 0},
      {"64 lanes, one address", ~0ull, 0, 1},
      {"32 lanes (every other), quads' halves share 64 B", every(2, 32), 2, 0},
      {"16 lanes (low quarter), one address", 0xFFFFull, 0, 1},
  };
  const size_t tables[] = {16u << 10, 2u << 20, 96u << 20, 1u << 30};
  const char* table_names[] = {"16 KB (L1)", "2 MB (L2)", "96 MB (Infinity Cache)", "1 GB (HBM)"};
  const int waves_per_cu_list[] = {20, 8};
  const bool quick = argc > 1 && !strcmp(argv[1], "quick");

  for (int wi = 0; wi < (quick ? 1 : 2); ++wi) {
    const int wpc = waves_per_cu_list[wi];
    // ballast so that exactly `wpc` one-wavefront blocks fit a CU's 160 KB of LDS
    const size_t lds = (160u * 1024u / wpc) & ~255u;
    for (int inflight = 1; inflight <= 4; inflight *= 4) {
      for (int ti = 0; ti < 4; ++ti) {
        const uint32_t n_mask = (uint32_t)(tables[ti] / 16u) - 1u;
        for (const Case& c : cases) {
          const uint32_t iters = ti == 3 ? 200 : 400;
          const uint32_t grid = cus * wpc * 4;
          float best = 1e30f;
          for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipEventRecord(e0, 0));
            if (inflight == 1)
              hipLaunchKernelGGL(gather_kernel<1>, dim3(grid), dim3(64), lds, 0, tab, n_mask, iters, c.lanes, c.share,
                                 c.same, sink);
            else
              hipLaunchKernelGGL(gather_kernel<4>, dim3(grid), dim3(64), lds, 0, tab, n_mask, iters, c.lanes, c.share,
                                 c.same, sink);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
          }
          const double insts = (double)grid * iters * inflight;
          const double cyc = best * 1e-3 * clock_hz * cus / insts;
          const int nl = __builtin_popcountll(c.lanes);
          printf(
              "{\"probe\": \"gather16\", \"waves_per_cu\": %d, \"inflight\": %d, \"table\": \"%s\", \"case\": \"%s\", "
              "\"lanes\": %d, \"ms\": %.4f, \"cu_cycles_per_wave_inst\": %.2f, \"cu_cycles_per_lane\": %.3f, "
              "\"gb_per_s\": %.1f}\n",
              wpc, inflight, table_names[ti], c.name, nl, best, cyc, cyc / nl, insts * nl * 16.0 / (best * 1e-3) / 1e9);
          fflush(stdout);
        }
      }
    }
    // narrower loads, every lane its own place
    for (int ti = 1; ti < 3; ++ti) {
      const uint32_t n_mask = (uint32_t)(tables[ti] / 16u) - 1u;
      for (int words = 1; words <= 3; ++words) {
        for (uint64_t lanes : {(uint64_t)~0ull, (uint64_t)0xFFFFull}) {
          const uint32_t iters = 400, grid = cus * wpc * 4;
          float best = 1e30f;
          for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipEventRecord(e0, 0));
            const uint32_t* t32 = reinterpret_cast<const uint32_t*>(tab);
            if (words == 1)
              hipLaunchKernelGGL(gather_narrow_kernel<1>, dim3(grid), dim3(64), lds, 0, t32, n_mask, iters, lanes, sink);
            else if (words == 2)
              hipLaunchKernelGGL(gather_narrow_kernel<2>, dim3(grid), dim3(64), lds, 0, t32, n_mask, iters, lanes, sink);
            else
              hipLaunchKernelGGL(gather_narrow_kernel<3>, dim3(grid), dim3(64), lds, 0, t32, n_mask, iters, lanes, sink);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
          }
          const double insts = (double)grid * iters;
          const double cyc = best * 1e-3 * clock_hz * cus / insts;
          const int nl = __builtin_popcountll(lanes);
          printf(
              "{\"probe\": \"gather%d\", \"waves_per_cu\": %d, \"inflight\": 1, \"table\": \"%s\", \"case\": \"%d lanes, own "
              "place\", \"lanes\": %d, \"ms\": %.4f, \"cu_cycles_per_wave_inst\": %.2f, \"cu_cycles_per_lane\": %.3f}\n",
              words * 4, wpc, table_names[ti], nl, nl, best, cyc, cyc / nl);
          fflush(stdout);
        }
      }
    }
  }

  // consecutive records of one place (MODE 1) and the neighbour one dependent step later (MODE 2)
  for (int wi = 0; wi < 3; ++wi) {
    const int wpc = wi == 0 ? 20 : (wi == 1 ? 8 : 4);
    const size_t lds = (160u * 1024u / wpc) & ~255u;
    for (int ti = 1; ti < 3; ++ti) {
      const uint32_t n_mask = (uint32_t)(tables[ti] / 16u) - 1u;
      for (uint64_t lanes : {(uint64_t)~0ull, every(3, 21), (uint64_t)0xFFull}) {
        for (int variant = 0; variant < 4; ++variant) {
          const uint32_t iters = 400, grid = cus * wpc * 4;
          float best = 1e30f;
          for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipEventRecord(e0, 0));
            if (variant == 0)
              hipLaunchKernelGGL((gather_mode_kernel<2, 1>), dim3(grid), dim3(64), lds, 0, tab, n_mask, iters, lanes, sink);
            else if (variant == 1)
              hipLaunchKernelGGL((gather_mode_kernel<4, 1>), dim3(grid), dim3(64), lds, 0, tab, n_mask, iters, lanes, sink);
            else if (variant == 2)
              hipLaunchKernelGGL((gather_mode_kernel<8, 1>), dim3(grid), dim3(64), lds, 0, tab, n_mask, iters, lanes, sink);
            else
              hipLaunchKernelGGL((gather_mode_kernel<1, 2>), dim3(grid), dim3(64), lds, 0, tab, n_mask, iters, lanes, sink);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
          }
          const int per_round = variant == 0 ? 2 : (variant == 1 ? 4 : (variant == 2 ? 8 : 2));
          const double rounds = (double)grid * iters;
          const double cyc_round = best * 1e-3 * clock_hz * cus / rounds;
          const int nl = __builtin_popcountll(lanes);
          printf(
              "{\"probe\": \"%s\", \"waves_per_cu\": %d, \"table\": \"%s\", \"lanes\": %d, \"loads_per_round\": %d, \"ms\": "
              "%.4f, \"cu_cycles_per_round\": %.2f, \"cu_cycles_per_wave_inst\": %.2f, \"cu_cycles_per_lane_line\": %.3f}\n",
              variant < 3 ? "consecutive records of one line" : "neighbour record one dependent step later", wpc,
              table_names[ti], nl, per_round, best, cyc_round, cyc_round / per_round, cyc_round / nl);
          fflush(stdout);
        }
      }
    }
  }
  fprintf(stderr, "device %s, %d CUs, %.0f MHz\n", prop.name, cus, clock_hz / 1e6);
  return 0;
}
