// Dev probe (round 5): CU-masked streams (hipExtStreamCreateWithCUMask) on MI355X.
//   1. which XCCs / CUs does a mask bit select?  (bit b -> XCC b % 8 if the driver interleaves the mask over the XCCs)
//   2. does a kernel node CAPTURED from a masked stream keep its mask when the graph is replayed (on the same stream, on another)?
//   3. what does a streaming kernel and a compute kernel lose on 2 / 4 / 6 of the 8 XCCs, alone and beside a second kernel on the
//      complementary XCCs (spatial partition) against the same two kernels time-sharing every CU?
// Every work-group records (XCC_ID, HW_ID); the host prints the set of XCCs and the number of distinct (XCC, SE, SH, CU) seen.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <set>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void where_k(uint32_t* out, int spin) {
    if (threadIdx.x == 0) {
        const uint32_t xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf;         // HW_REG_XCC_ID
        const uint32_t hw = __builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4);     // HW_REG_HW_ID bits 15:0 (cu_id 11:8, sh 12, se 15:13)
        out[blockIdx.x] = (xcc << 16) | (hw & 0xffff);
    }
    float a = threadIdx.x;
    for (int i = 0; i < spin; ++i) a = a * 1.0001f + 0.5f;
    if (a == 12345.f) out[0] = 0;
}
// streaming: y = x + 1 over n float4
__global__ void stream_k(const float4* __restrict__ x, float4* __restrict__ y, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float4 v = x[i]; v.x += 1.f; v.y += 1.f; v.z += 1.f; v.w += 1.f; y[i] = v;
    }
}
// compute: register-resident FMA chain (no memory)
__global__ void fma_k(float* out, int iters) {
    float a = threadIdx.x * 1e-3f, b = 1.0001f, c = 0.5f, d = a + 1.f, e = a + 2.f, f = a + 3.f;
    for (int i = 0; i < iters; ++i) { a = a * b + c; d = d * b + c; e = e * b + c; f = f * b + c; }
    if (a + d + e + f == 12345.f) out[0] = a;
}

static void report(const char* what, const std::vector<uint32_t>& h) {
    std::set<uint32_t> xccs, cus;
    for (uint32_t v : h) { xccs.insert(v >> 16); cus.insert(((v >> 16) << 16) | (v & 0xff00)); }
    printf("%-58s XCCs {", what);
    for (uint32_t x : xccs) printf(" %u", x);
    printf(" }  distinct (xcc, se, sh, cu): %zu\n", cus.size());
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s, %d CUs\n", prop.name, prop.multiProcessorCount);
    const int NB = 4096;
    uint32_t* d;
    CK(hipMalloc((void**)&d, NB * 4));
    std::vector<uint32_t> h(NB);
    auto run_where = [&](hipStream_t s, const char* what) -> int {
        CK(hipMemsetAsync(d, 0xff, NB * 4, s));
        hipLaunchKernelGGL(where_k, dim3(NB), dim3(256), 0, s, d, 2000);
        CK(hipStreamSynchronize(s));
        CK(hipMemcpy(h.data(), d, NB * 4, hipMemcpyDeviceToHost));
        report(what, h);
        return 0;
    };
    hipStream_t plain;
    CK(hipStreamCreate(&plain));
    if (run_where(plain, "plain stream:")) return 1;

    // --- 1. mask layouts: 256 bits
    auto make_mask = [](auto pred) { std::vector<uint32_t> m(8, 0u); for (int b = 0; b < 256; ++b) if (pred(b)) m[b >> 5] |= 1u << (b & 31); return m; };
    // (first run of this probe: bit b selects CU b / 8 of XCC b % 8; a mask that leaves an XCC without any CU is ignored for that XCC -
    // "b % 8 < 2" ran on all 256 CUs - so the partition is by CU index inside EVERY XCC, not by XCC)
    struct { const char* name; std::vector<uint32_t> m; } masks[] = {
        {"mask bits 0..63 (CUs 0-7 of every XCC):", make_mask([](int b) { return b < 64; })},
        {"mask bits 0..63 again (a second stream, same mask):", make_mask([](int b) { return b < 64; })},
        {"mask bits 64..255 (CUs 8-31 of every XCC):", make_mask([](int b) { return b >= 64; })},
        {"mask bits 0..127 (CUs 0-15 of every XCC):", make_mask([](int b) { return b < 128; })},
    };
    hipStream_t ms[4];
    for (int i = 0; i < 4; ++i) {
        CK(hipExtStreamCreateWithCUMask(&ms[i], 8, masks[i].m.data()));
        if (run_where(ms[i], masks[i].name)) return 1;
    }

    // --- 2. capture from the masked stream (XCC 0,1), replay
    {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(ms[1], hipStreamCaptureModeGlobal));
        hipLaunchKernelGGL(where_k, dim3(NB), dim3(256), 0, ms[1], d, 2000);
        CK(hipStreamEndCapture(ms[1], &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int k = 0; k < 2; ++k) {
            hipStream_t rs = k ? plain : ms[1];
            CK(hipMemsetAsync(d, 0xff, NB * 4, rs));
            CK(hipGraphLaunch(ge, rs));
            CK(hipStreamSynchronize(rs));
            CK(hipMemcpy(h.data(), d, NB * 4, hipMemcpyDeviceToHost));
            report(k ? "graph captured on the 64-CU stream, replayed on PLAIN:" : "graph captured on the 64-CU stream, replayed on it:", h);
        }
        // fork inside a capture that starts on the plain stream: plain -> event -> masked stream -> kernel -> join
        hipEvent_t e0, e1;
        CK(hipEventCreateWithFlags(&e0, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&e1, hipEventDisableTiming));
        CK(hipStreamBeginCapture(plain, hipStreamCaptureModeGlobal));
        CK(hipEventRecord(e0, plain));
        CK(hipStreamWaitEvent(ms[1], e0, 0));
        hipLaunchKernelGGL(where_k, dim3(NB), dim3(256), 0, ms[1], d, 2000);
        CK(hipEventRecord(e1, ms[1]));
        CK(hipStreamWaitEvent(plain, e1, 0));
        hipGraph_t g2; hipGraphExec_t ge2;
        CK(hipStreamEndCapture(plain, &g2));
        CK(hipGraphInstantiate(&ge2, g2, nullptr, nullptr, 0));
        CK(hipMemset(d, 0xff, NB * 4));
        CK(hipGraphLaunch(ge2, plain));
        CK(hipStreamSynchronize(plain));
        CK(hipMemcpy(h.data(), d, NB * 4, hipMemcpyDeviceToHost));
        report("capture on PLAIN, kernel forked to the 64-CU stream, replay:", h);
    }

    // --- 3. time: streaming and compute kernels, full chip | k XCCs | spatial split vs time sharing
    const size_t n4 = (size_t)64 << 20;          // 1 GiB in + 1 GiB out
    float4 *x, *y, *x2, *y2; float* fo;
    CK(hipMalloc((void**)&x, n4 * 16)); CK(hipMalloc((void**)&y, n4 * 16)); CK(hipMalloc((void**)&x2, n4 * 16)); CK(hipMalloc((void**)&y2, n4 * 16));
    CK(hipMalloc((void**)&fo, 4096));
    CK(hipMemset(x, 0, n4 * 16)); CK(hipMemset(x2, 0, n4 * 16));
    hipEvent_t t0, t1;
    CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    auto time_pair = [&](hipStream_t sa, hipStream_t sb, int kind_a, int kind_b, const char* what) -> int {
        // kind: 0 none, 1 streaming (2 GiB of traffic), 2 compute
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(t0, plain));
            CK(hipStreamWaitEvent(sa, t0, 0)); if (kind_b) CK(hipStreamWaitEvent(sb, t0, 0));
            if (kind_a == 1) hipLaunchKernelGGL(stream_k, dim3(8192), dim3(256), 0, sa, x, y, n4);
            if (kind_a == 2) hipLaunchKernelGGL(fma_k, dim3(4096), dim3(256), 0, sa, fo, 200000);
            if (kind_b == 1) hipLaunchKernelGGL(stream_k, dim3(8192), dim3(256), 0, sb, x2, y2, n4);
            if (kind_b == 2) hipLaunchKernelGGL(fma_k, dim3(4096), dim3(256), 0, sb, fo + 512, 200000);
            hipEvent_t ea, eb;
            CK(hipEventCreateWithFlags(&ea, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&eb, hipEventDisableTiming));
            CK(hipEventRecord(ea, sa)); CK(hipStreamWaitEvent(plain, ea, 0));
            if (kind_b) { CK(hipEventRecord(eb, sb)); CK(hipStreamWaitEvent(plain, eb, 0)); }
            CK(hipEventRecord(t1, plain));
            CK(hipEventSynchronize(t1));
            float msv; CK(hipEventElapsedTime(&msv, t0, t1));
            if (msv < best) best = msv;
            CK(hipEventDestroy(ea)); CK(hipEventDestroy(eb));
        }
        printf("%-78s %.3f ms\n", what, best);
        return 0;
    };
    hipStream_t plain2;
    CK(hipStreamCreate(&plain2));
    if (time_pair(plain, plain, 1, 0, "streaming 2 GiB, full chip:")) return 1;
    if (time_pair(ms[1], ms[1], 1, 0, "streaming 2 GiB, 64 CUs:")) return 1;
    if (time_pair(ms[3], ms[3], 1, 0, "streaming 2 GiB, 128 CUs:")) return 1;
    if (time_pair(ms[2], ms[2], 1, 0, "streaming 2 GiB, 192 CUs:")) return 1;
    if (time_pair(plain, plain, 2, 0, "compute, full chip:")) return 1;
    if (time_pair(ms[1], ms[1], 2, 0, "compute, 64 CUs:")) return 1;
    if (time_pair(ms[2], ms[2], 2, 0, "compute, 192 CUs:")) return 1;
    if (time_pair(plain, plain2, 2, 1, "compute + streaming, both on every CU (time sharing):")) return 1;
    if (time_pair(ms[2], ms[1], 2, 1, "compute on 192 CUs + streaming on 64 CUs (spatial):")) return 1;
    if (time_pair(ms[1], ms[2], 2, 1, "compute on 64 CUs + streaming on 192 CUs (spatial):")) return 1;
    if (time_pair(plain, plain2, 1, 1, "streaming + streaming, time sharing:")) return 1;
    if (time_pair(ms[2], ms[1], 1, 1, "streaming on 192 + streaming on 64 CUs:")) return 1;
    printf("done\n");
    return 0;
}
