// xcd_handoff_bench.hip -- what does a hand-off between workgroups of the SAME XCD cost on gfx950, by store / load flavour?
// (measurement tool for DESIGN.md's hierarchical gather; not part of the engine)
//   hipcc --offload-arch=gfx950 -O3 -o build/xcd_handoff_bench tools/micro/xcd_handoff_bench.hip && build/xcd_handoff_bench
// 256 workgroups x 256 threads, one per CU (148 KB of LDS each).  Every workgroup reads its XCC id, the ids are all-gathered
// once, and then every XCD's members run ROUNDS all-gathers among THEMSELVES: member m publishes a tagged 8-byte granule,
// every member polls all members' granules of that round.  Flavours:
//   0: agent-scope relaxed atomic store + load (sc1 both sides: write-through, served by the fabric)          -- the flat protocol's
//   1: plain store + agent-scope (sc1) load
//   2: workgroup-scope atomic exchange + workgroup-scope atomic fetch_or(0)   (both executed by the XCD's L2)
//   3: plain store + workgroup-scope atomic fetch_or(0)
//   4: the same all-gather over ALL workgroups with flavour 0 (the chip-wide hop, for comparison)
//   5: two levels: flavour 1 inside the XCD, then the first member of every XCD publishes ONE granule chip-wide (sc1) and every
//      workgroup polls the <= 16 group granules (sc1)
//   6: only the second level of 5 (one publisher per XCD, every workgroup polls the group granules)
// Reports per flavour: cycles per round (s_memtime, workgroup 0 of each XCD), rounds that timed out (stale / never visible).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned long long u64;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
#define AG_LOAD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define AG_STORE(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)

// exact instruction forms (the compiler turns a workgroup-scope fetch_or(p, 0) into a plain sc0 LOAD and a volatile store into
// a system-scope one -- neither is what is being measured)
__device__ __forceinline__ void plain_store(u64* p, u64 v) { asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void l2_swap(u64* p, u64 v) { asm volatile("global_atomic_swap_x2 %0, %1, off" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ u64 l2_read(u64* p) {  // returning atomic OR with 0: executed by the L2, never by the L1
    u64 r;
    const u64 z = 0;
    asm volatile("global_atomic_or_x2 %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p), "v"(z) : "memory");
    return r;
}

struct Args {
    u64* xcctab;   // [G]
    u64* gran;     // [2][G]
    u64* grp;      // [2][16] group granules (flavours 5, 6)
    u64* out;      // [G][4]: cycles, timeouts, xcc, n_mem
    int G, rounds, flavour, stride;  // stride: u64 words between consecutive workgroups' granules (1 = packed; 512 = one per 4 KB)
    unsigned spin_limit;
};

__global__ void __launch_bounds__(256) k_bench(Args a) {
    extern __shared__ unsigned char lds_raw[];
    int* xcc_of = reinterpret_cast<int*>(lds_raw);
    int* mem = xcc_of + 256;
    __shared__ int n_mem, bad;
    __shared__ int grp_used[16];
    const int tid = threadIdx.x, b = blockIdx.x;
    const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 0xfu;  // HW_REG_XCC_ID[3:0]
    if (tid == 0) { AG_STORE(a.xcctab + b, (1ull << 32) | xcc); bad = 0; }
    if (tid < a.G) {
        unsigned spins = 0;
        for (;;) {
            const u64 x = AG_LOAD(a.xcctab + tid);
            if ((x >> 32) == 1ull) { xcc_of[tid] = (int)(x & 0xf); break; }
            __builtin_amdgcn_s_sleep(2);
            if (++spins > a.spin_limit) { xcc_of[tid] = -1; break; }
        }
    }
    __syncthreads();
    if (tid == 0) {
        int n = 0;
        for (int w = 0; w < a.G; w++)
            if (a.flavour == 4 || xcc_of[w] == (int)xcc) mem[n++] = w;  // (flavours 5, 6: my XCD's members, like 0..3)
        n_mem = a.flavour == 6 ? 0 : n;
        if (a.flavour == 6) mem[0] = n ? mem[0] : -1;
        for (int x = 0; x < 16; x++) grp_used[x] = 0;
        for (int w = 0; w < a.G; w++) if (xcc_of[w] >= 0) grp_used[xcc_of[w]] = 1;
    }
    __syncthreads();
    const int nm = n_mem;
    u64 timeouts = 0;
    const u64 t0 = __builtin_amdgcn_s_memtime();
    for (int round = 0; round < a.rounds; round++) {
        const int par = round & 1;
        const u64 tag = (u64)(round + 1);
        u64* mine = a.gran + ((size_t)par * a.G + b) * a.stride;
        const u64 val = (tag << 32) | (unsigned)(b * 7 + round);
        if (tid == 0) {
            switch (a.flavour) {
                case 0: case 4: AG_STORE(mine, val); break;
                case 1: case 3: case 5: plain_store(mine, val); break;
                case 6: break;
                case 2: l2_swap(mine, val); break;
            }
        }
        int ok = 1;
        for (int base = 0; base < nm; base += 256) {
            const int k = base + tid;
            if (k < nm) {
                u64* p = a.gran + ((size_t)par * a.G + mem[k]) * a.stride;
                unsigned spins = 0;
                for (;;) {
                    u64 x;
                    if (a.flavour == 2 || a.flavour == 3) x = l2_read(p);
                    else x = AG_LOAD(p);
                    if ((x >> 32) == tag) { if ((unsigned)x != (unsigned)(mem[k] * 7 + round)) ok = 0; break; }
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > a.spin_limit) { ok = 0; break; }
                }
            }
        }
        if (a.flavour >= 5) {
            __syncthreads();
            // second level: the first member of my XCD publishes, everybody polls every XCD's granule
            if (tid == 0 && mem[0] == b) AG_STORE(a.grp + par * 16 + xcc, (tag << 32) | (unsigned)(xcc * 11 + round));
            if (tid < 16 && grp_used[tid]) {
                u64* p = a.grp + par * 16 + tid;
                unsigned spins = 0;
                for (;;) {
                    const u64 x = AG_LOAD(p);
                    if ((x >> 32) == tag) { if ((unsigned)x != (unsigned)(tid * 11 + round)) ok = 0; break; }
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > a.spin_limit) { ok = 0; break; }
                }
            }
        }
        if (!__syncthreads_and(ok)) timeouts++;
    }
    const u64 t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) {
        a.out[b * 4 + 0] = (t1 - t0) / (u64)a.rounds;
        a.out[b * 4 + 1] = timeouts;
        a.out[b * 4 + 2] = xcc;
        a.out[b * 4 + 3] = (u64)nm;
    }
}

int main(int argc, char** argv) {
    const int G = argc > 1 ? atoi(argv[1]) : 256, rounds = argc > 2 ? atoi(argv[2]) : 2000, stride = argc > 3 ? atoi(argv[3]) : 1;
    u64 *xcctab, *gran, *out, *grp;
    CHECK(hipMalloc(&grp, sizeof(u64) * 32));
    CHECK(hipMalloc(&xcctab, sizeof(u64) * G));
    CHECK(hipMalloc(&gran, sizeof(u64) * 2 * G * stride));
    CHECK(hipMalloc(&out, sizeof(u64) * 4 * G));
    const size_t lds = 148 * 1024;  // one workgroup per CU
    CHECK(hipFuncSetAttribute((const void*)k_bench, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const char* names[] = {"sc1 store + sc1 load (agent atomics)", "plain store + sc1 load", "wg-scope atomic xchg + wg-scope fetch_or(0)",
                           "plain store + wg-scope fetch_or(0)", "chip-wide all-gather, sc1 both sides",
                           "two levels: plain + sc1 inside the XCD, then one sc1 granule per XCD", "second level alone"};
    for (int fl = 0; fl < 7; fl++) {
        if (fl == 2 || fl == 3) continue;  // (L2-executed atomics: measured 8.8-12.7 k cycles per round, r03_b)
        for (int rep = 0; rep < 2; rep++) {
            CHECK(hipMemset(xcctab, 0, sizeof(u64) * G));
            CHECK(hipMemset(gran, 0, sizeof(u64) * 2 * G * stride));
            CHECK(hipMemset(out, 0, sizeof(u64) * 4 * G));
            CHECK(hipMemset(grp, 0, sizeof(u64) * 32));
            Args a{xcctab, gran, grp, out, G, rounds, fl, stride, 1u << 16};
            void* args[] = {&a};
            CHECK(hipLaunchCooperativeKernel((const void*)k_bench, dim3(G), dim3(256), args, lds, 0));
            CHECK(hipDeviceSynchronize());
            std::vector<u64> h(4 * G);
            CHECK(hipMemcpy(h.data(), out, sizeof(u64) * 4 * G, hipMemcpyDeviceToHost));
            u64 cyc_min = ~0ull, cyc_max = 0, to = 0;
            int members[16] = {0};
            for (int b = 0; b < G; b++) {
                cyc_min = h[b * 4] < cyc_min ? h[b * 4] : cyc_min;
                cyc_max = h[b * 4] > cyc_max ? h[b * 4] : cyc_max;
                to += h[b * 4 + 1];
                members[h[b * 4 + 2] & 15]++;
            }
            if (rep == 1) {
                printf("stride %d words, flavour %d (%s): cycles/round min %llu max %llu, timed-out rounds (summed over workgroups) %llu of %d x %d; members per XCC:", stride, fl, names[fl],
                       cyc_min, cyc_max, to, rounds, G);
                for (int x = 0; x < 16; x++) if (members[x]) printf(" %d:%d", x, members[x]);
                printf("\n");
            }
        }
    }
    // block -> XCC map as observed
    {
        std::vector<u64> h(4 * G);
        CHECK(hipMemcpy(h.data(), out, sizeof(u64) * 4 * G, hipMemcpyDeviceToHost));
        printf("xcc of blocks 0..15:");
        for (int b = 0; b < 16 && b < G; b++) printf(" %llu", h[b * 4 + 2]);
        printf("\n");
    }
    return 0;
}
