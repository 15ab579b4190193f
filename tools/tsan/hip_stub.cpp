// A HIP runtime that has no device: the 31 entry points libpsk.so imports, for tools/tsan_host.sh (the
// threaded host ingest -- reader + upload | inflate | count -- had no race-detector pass; GPU sanitizers are not available on
// this pool and TSan cannot follow a real runtime's threads anyway).  "Device" memory is host memory, copies and memsets are
// done on the spot, kernel launches do NOTHING and succeed: the host code's threads, locks and hand-overs run exactly as they do
// in the product; what the kernels would have computed is zeros (so the device inflate declines every file and zlib inflates
// it on the pool's threads, and every sample counts zero k-mers).  Test infrastructure; never linked into the product.
#include <hip/hip_runtime_api.h>

#include <cstdlib>
#include <cstring>

extern "C" {

struct StubConfig {
    dim3 grid, block;
    size_t shmem;
    hipStream_t stream;
};
static thread_local StubConfig g_cfg;
static void *g_handle[1];

void **__hipRegisterFatBinary(const void *) { return g_handle; }
void __hipRegisterFunction(void **, const void *, char *, const char *, unsigned, void *, void *, void *, void *, int *) {}
void __hipUnregisterFatBinary(void **) {}
hipError_t __hipPushCallConfiguration(dim3 grid, dim3 block, size_t shmem, hipStream_t stream)
{
    g_cfg = StubConfig{grid, block, shmem, stream};
    return hipSuccess;
}
hipError_t __hipPopCallConfiguration(dim3 *grid, dim3 *block, size_t *shmem, hipStream_t *stream)
{
    *grid = g_cfg.grid;
    *block = g_cfg.block;
    *shmem = g_cfg.shmem;
    *stream = g_cfg.stream;
    return hipSuccess;
}
hipError_t hipLaunchKernel(const void *, dim3, dim3, void **, size_t, hipStream_t) { return hipSuccess; }

hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetDevicePropertiesR0600(hipDeviceProp_tR0600 *p, int)
{
    std::memset(p, 0, sizeof *p);
    std::strcpy(p->name, "no device (tools/tsan/hip_stub.cpp)");
    std::strcpy(p->gcnArchName, "gfx950");
    p->multiProcessorCount = 256;
    p->totalGlobalMem = (size_t)64 << 30;
    p->warpSize = 64;
    return hipSuccess;
}
hipError_t hipMemGetInfo(size_t *free_b, size_t *total_b) { *free_b = (size_t)48 << 30; *total_b = (size_t)64 << 30; return hipSuccess; }
const char *hipGetErrorString(hipError_t) { return "stub"; }
hipError_t hipGetLastError(void) { return hipSuccess; }
hipError_t hipFuncSetAttribute(const void *, hipFuncAttribute, int) { return hipSuccess; }

hipError_t hipMalloc(void **p, size_t n) { *p = std::calloc(n ? n : 1, 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void *p) { std::free(p); return hipSuccess; }
hipError_t hipHostMalloc(void **p, size_t n, unsigned) { *p = std::calloc(n ? n : 1, 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void *p) { std::free(p); return hipSuccess; }
hipError_t hipHostGetDevicePointer(void **d, void *h, unsigned) { *d = h; return hipSuccess; }
hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { if (n) std::memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { if (n) std::memmove(d, s, n); return hipSuccess; }
hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { if (n) std::memset(d, v, n); return hipSuccess; }

hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = reinterpret_cast<hipStream_t>(std::malloc(8)); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { std::free(s); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t *e) { *e = reinterpret_cast<hipEvent_t>(std::malloc(8)); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = reinterpret_cast<hipEvent_t>(std::malloc(8)); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { std::free(e); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0.001f; return hipSuccess; }
}
