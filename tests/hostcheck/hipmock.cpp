// TEST INFRASTRUCTURE. A stand-in for the slice of the HIP runtime that ppde_amd/csrc/ppde_api.hip calls, on host
// memory, so that the HOST side of the C ABI (allocation bookkeeping, copies and their sizes, error paths, stream /
// event / graph object lifetimes) can run under AddressSanitizer + LeakSanitizer in a container without a GPU.
// Kernels are not executed (hipLaunchKernel returns success); "device" memory is calloc'd host memory, so every
// hipMemcpy / hipMemset the host layer issues is bounds-checked by the sanitizer against the matching hipMalloc.
// Every stream, event, graph and executable graph is its own heap object: one the library forgets to destroy
// shows up in the leak report.
// Fault injection: HIPMOCK_FAIL_AT=k makes the k-th fallible call (allocations, stream / event / graph creation)
// return an error, which drives the library's clean-up paths; hipmock_calls() reports how many there were.
#include <hip/hip_runtime_api.h>
#include <cstdlib>
#include <cstring>

namespace {
long g_calls = 0, g_fail_at = -1;
bool g_armed = false;
bool fail_now() {
    if (!g_armed) { const char* e = getenv("HIPMOCK_FAIL_AT"); g_fail_at = e ? atol(e) : -1; g_armed = true; }
    return ++g_calls == g_fail_at;
}
struct Cfg { dim3 grid, block; size_t shmem; hipStream_t stream; };
thread_local Cfg t_cfg;
}  // namespace

extern "C" {
long hipmock_calls() { return g_calls; }
void hipmock_rearm(long fail_at) { g_calls = 0; g_fail_at = fail_at; g_armed = true; }

void** __hipRegisterFatBinary(const void*) { static void* h; return &h; }
void __hipUnregisterFatBinary(void**) {}
void __hipRegisterFunction(void**, const void*, char*, const char*, unsigned, void*, void*, void*, void*, int*) {}
void __hipRegisterVar(void**, void*, char*, const char*, int, size_t, int, int) {}
hipError_t __hipPushCallConfiguration(dim3 g, dim3 b, size_t sh, hipStream_t s) { t_cfg = Cfg{g, b, sh, s}; return hipSuccess; }
hipError_t __hipPopCallConfiguration(dim3* g, dim3* b, size_t* sh, hipStream_t* s) { *g = t_cfg.grid; *b = t_cfg.block; *sh = t_cfg.shmem; *s = t_cfg.stream; return hipSuccess; }
hipError_t hipLaunchKernel(const void*, dim3 g, dim3 b, void**, size_t sh, hipStream_t) {
    if (g.x == 0 || g.y == 0 || g.z == 0 || b.x == 0 || b.x * b.y * b.z > 1024 || sh > 160 * 1024) return hipErrorInvalidConfiguration;
    return hipSuccess;
}
hipError_t hipExtLaunchKernel(const void* f, dim3 g, dim3 b, void** args, size_t sh, hipStream_t st, hipEvent_t, hipEvent_t, int) {
    return hipLaunchKernel(f, g, b, args, sh, st);
}
hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipSetDevice(int d) { return d == 0 ? hipSuccess : hipErrorInvalidDevice; }
hipError_t hipGetLastError() { return hipSuccess; }
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : e == hipErrorOutOfMemory ? "out of memory (mock)" : "error (mock)"; }
hipError_t hipDeviceSynchronize() { return hipSuccess; }

hipError_t hipMalloc(void** p, size_t n) { if (fail_now()) { *p = nullptr; return hipErrorOutOfMemory; } *p = calloc(n ? n : 1, 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void* p) { free(p); return hipSuccess; }
hipError_t hipHostMalloc(void** p, size_t n, unsigned) { if (fail_now()) { *p = nullptr; return hipErrorOutOfMemory; } *p = calloc(n ? n : 1, 1); return hipSuccess; }
hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
hipError_t hipHostGetDevicePointer(void** d, void* h, unsigned) { *d = h; return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { if (n) memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { if (n) memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemset(void* d, int v, size_t n) { if (n) memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { if (n) memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetD32Async(hipDeviceptr_t d, int v, size_t count, hipStream_t) { for (size_t i = 0; i < count; ++i) ((int*)d)[i] = v; return hipSuccess; }

hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { if (fail_now()) return hipErrorOutOfMemory; *s = (hipStream_t)malloc(8); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { free((void*)s); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { if (fail_now()) return hipErrorOutOfMemory; *e = (hipEvent_t)malloc(8); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e) { free((void*)e); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 1.0f; return hipSuccess; }
hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { return fail_now() ? hipErrorOutOfMemory : hipSuccess; }
hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* g) { if (fail_now()) { *g = nullptr; return hipErrorOutOfMemory; } *g = (hipGraph_t)malloc(8); return hipSuccess; }
hipError_t hipGraphInstantiate(hipGraphExec_t* x, hipGraph_t, hipGraphNode_t*, char*, size_t) { if (fail_now()) return hipErrorOutOfMemory; *x = (hipGraphExec_t)malloc(8); return hipSuccess; }
hipError_t hipGraphUpload(hipGraphExec_t, hipStream_t) { return hipSuccess; }
hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return hipSuccess; }
hipError_t hipGraphDestroy(hipGraph_t g) { free((void*)g); return hipSuccess; }
hipError_t hipGraphExecDestroy(hipGraphExec_t x) { free((void*)x); return hipSuccess; }
}
