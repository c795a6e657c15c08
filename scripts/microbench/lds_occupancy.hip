// Measured workgroup residency per CU on gfx950 as a function of workgroup size and LDS bytes.
// Every workgroup records the CU it ran on (HW_ID / XCC_ID) and its start / end time
// (s_memrealtime); the host counts the largest number of workgroups alive at once on one CU and
// prints it next to hipOccupancyMaxActiveBlocksPerMultiprocessor.
//   hipcc --offload-arch=gfx950 -O2 scripts/microbench/lds_occupancy.hip -o build/lds_occupancy
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <map>
#include <vector>
struct Rec {
    unsigned hwid, xcc;
    unsigned long long t0, t1;
};
__global__ void k(Rec *out, int spin, int ldswords)
{
    extern __shared__ double s[];
    const unsigned long long t0 = __builtin_readcyclecounter();
    const unsigned long long r0 = wall_clock64();
    s[threadIdx.x % ldswords] = threadIdx.x;
    __syncthreads();
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(64);
    __syncthreads();
    if (threadIdx.x == 0) {
        Rec r;
        r.hwid = __builtin_amdgcn_s_getreg((31 << 11) | 4);
        r.xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        r.t0 = r0;
        r.t1 = wall_clock64();
        out[blockIdx.x] = r;
        if (t0 == 1) out[0].hwid = (unsigned)s[1];
    }
}
// The same probe with a forced VGPR allocation (clobbering the highest register): does a second
// 6-wave workgroup become resident when the register file only leaves 3 waves per SIMD?
template <int VG>
__global__ void __launch_bounds__(384) kv(Rec *out, unsigned *simd, int spin, int ldswords)
{
    extern __shared__ double s[];
    const unsigned long long r0 = wall_clock64();
    if (VG == 128) asm volatile("v_mov_b32 v127, 0" ::: "v127");
    if (VG == 168) asm volatile("v_mov_b32 v167, 0" ::: "v167");
    if (VG == 96) asm volatile("v_mov_b32 v95, 0" ::: "v95");
    s[threadIdx.x % ldswords] = threadIdx.x;
    __syncthreads();
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(64);
    __syncthreads();
    if (threadIdx.x % 64 == 0)
        simd[blockIdx.x * 8 + threadIdx.x / 64] = (__builtin_amdgcn_s_getreg((31 << 11) | 4) >> 4) & 3;
    if (threadIdx.x == 0) {
        Rec r;
        r.hwid = __builtin_amdgcn_s_getreg((31 << 11) | 4);
        r.xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        r.t0 = r0;
        r.t1 = wall_clock64();
        out[blockIdx.x] = r;
    }
}
static int resident(std::vector<Rec> &h)
{
    std::map<unsigned long long, std::vector<std::pair<unsigned long long, int>>> ev;
    for (auto &r : h) {
        const unsigned long long key = ((unsigned long long)(r.xcc & 0xf) << 32) | (r.hwid & 0x0000ff00u);
        ev[key].push_back({r.t0, +1});
        ev[key].push_back({r.t1, -1});
    }
    int best = 0;
    for (auto &kv : ev) {
        auto &v = kv.second;
        std::sort(v.begin(), v.end(), [](auto &a, auto &b) { return a.first < b.first || (a.first == b.first && a.second < b.second); });
        int cur = 0;
        for (auto &e : v) {
            cur += e.second;
            best = std::max(best, cur);
        }
    }
    return best;
}
template <int VG>
void run_vgpr(Rec *d, unsigned *dsimd, int nblk, int nt, int lds)
{
    std::vector<Rec> h(nblk);
    std::vector<unsigned> hs(nblk * 8);
    hipFuncSetAttribute((const void *)kv<VG>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    int api = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&api, kv<VG>, nt, lds);
    kv<VG><<<nblk, nt, lds>>>(d, dsimd, 20, lds / 8);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), d, sizeof(Rec) * nblk, hipMemcpyDeviceToHost);
    hipMemcpy(hs.data(), dsimd, 4 * nblk * 8, hipMemcpyDeviceToHost);
    std::map<unsigned, int> pat;  // waves-per-SIMD pattern of a workgroup, e.g. 2211
    const int nw = (nt + 63) / 64;
    for (int b = 0; b < nblk; ++b) {
        int c[4] = {0, 0, 0, 0};
        for (int w = 0; w < nw; ++w) c[hs[b * 8 + w] & 3]++;
        pat[c[0] * 1000 + c[1] * 100 + c[2] * 10 + c[3]]++;
    }
    printf("{\"threads\": %d, \"lds_bytes\": %d, \"vgprs\": %d, \"occupancy_api\": %d, \"resident_measured\": %d, \"waves_per_simd_patterns\": {",
           nt, lds, VG, api, resident(h));
    bool first = true;
    for (auto &kv : pat) {
        printf("%s\"%04u\": %d", first ? "" : ", ", kv.first, kv.second);
        first = false;
    }
    printf("}}\n");
}
int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("{\"sharedMemPerBlock\": %zu, \"maxSharedMemoryPerMultiProcessor\": %zu, \"multiProcessorCount\": %d, "
           "\"regsPerMultiprocessor\": %d, \"maxThreadsPerMultiProcessor\": %d}\n",
           p.sharedMemPerBlock, p.maxSharedMemoryPerMultiProcessor, p.multiProcessorCount,
           p.regsPerMultiprocessor, p.maxThreadsPerMultiProcessor);
    const int nblk = 256 * 16;
    Rec *d;
    hipMalloc(&d, sizeof(Rec) * nblk);
    std::vector<Rec> h(nblk);
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int threads[] = {64, 192, 256, 343, 384, 512};
    const int lds[] = {1024, 16384, 31192, 32768, 40000, 49152, 54000, 65536, 81672, 81920, 106360, 163840};
    for (int nt : threads)
        for (int b : lds) {
            int api = 0;
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&api, k, nt, b);
            hipMemset(d, 0, sizeof(Rec) * nblk);
            k<<<nblk, nt, b>>>(d, 20, b / 8);
            if (hipDeviceSynchronize() != hipSuccess) {
                printf("{\"threads\": %d, \"lds_bytes\": %d, \"error\": \"%s\"}\n", nt, b, hipGetErrorString(hipGetLastError()));
                continue;
            }
            hipMemcpy(h.data(), d, sizeof(Rec) * nblk, hipMemcpyDeviceToHost);
            // CU key: XCC_ID + the cu_id [11:8], sh_id [12], se_id [15:13] bits of HW_ID
            std::map<unsigned long long, std::vector<std::pair<unsigned long long, int>>> ev;
            for (auto &r : h) {
                const unsigned long long key = ((unsigned long long)(r.xcc & 0xf) << 32) | (r.hwid & 0x0000ff00u);
                ev[key].push_back({r.t0, +1});
                ev[key].push_back({r.t1, -1});
            }
            int best = 0;
            for (auto &kv : ev) {
                auto &v = kv.second;
                std::sort(v.begin(), v.end(), [](auto &a, auto &b) { return a.first < b.first || (a.first == b.first && a.second < b.second); });
                int cur = 0;
                for (auto &e : v) {
                    cur += e.second;
                    best = std::max(best, cur);
                }
            }
            printf("{\"threads\": %d, \"lds_bytes\": %d, \"occupancy_api\": %d, \"resident_measured\": %d, \"cus_seen\": %zu}\n",
                   nt, b, api, best, ev.size());
        }
    unsigned *dsimd;
    hipMalloc(&dsimd, 4 * nblk * 8);
    for (int b : {31192, 54000, 65536, 81672}) {
        run_vgpr<96>(d, dsimd, nblk, 343, b);
        run_vgpr<128>(d, dsimd, nblk, 343, b);
        run_vgpr<168>(d, dsimd, nblk, 343, b);
    }
    run_vgpr<128>(d, dsimd, nblk, 192, 31192);
    run_vgpr<168>(d, dsimd, nblk, 192, 31192);
    return 0;
}
