// Error channel, version and device query of libdir_hip.so.
#include "dir_common.h"

#include <string.h>

namespace dir {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace dir

namespace dir {
long long* stamps_begin(const char* kernel) {
    const char* e = getenv("DIR_STAMPS");
    if (!e || strcmp(e, kernel) != 0) return nullptr;
    long long* buf = nullptr;
    if (hipMalloc((void**)&buf, MAX_STAMPS * sizeof(long long)) != hipSuccess) return nullptr;
    if (hipMemset(buf, 0, MAX_STAMPS * sizeof(long long)) != hipSuccess) { (void)hipFree(buf); return nullptr; }
    return buf;
}
void stamps_end(const char* kernel, long long* buf, hipStream_t s) {
    if (!buf) return;
    long long h[MAX_STAMPS] = {0};
    (void)hipStreamSynchronize(s);
    (void)hipMemcpy(h, buf, sizeof(h), hipMemcpyDeviceToHost);
    (void)hipFree(buf);
    fprintf(stderr, "%s stamps (ticks since start):", kernel);
    for (int i = 1; i < MAX_STAMPS && h[i]; ++i) fprintf(stderr, " %lld", h[i] - h[0]);
    fprintf(stderr, "\n");
}
}  // namespace dir

extern "C" int dir_abi_version(void) { return DIR_ABI_VERSION; }

extern "C" const char* dir_last_error(void) { return dir::g_err; }

extern "C" int dir_device_info(char* arch_host, int arch_len, int* num_cu_host) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        dir::set_error("no HIP device visible");
        return DIR_E_NODEVICE;
    }
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, 0) != hipSuccess) {
        dir::set_error("hipGetDeviceProperties failed");
        return DIR_E_LAUNCH;
    }
    if (arch_host && arch_len > 0) {
        strncpy(arch_host, p.gcnArchName, arch_len - 1);
        arch_host[arch_len - 1] = 0;
    }
    if (num_cu_host) *num_cu_host = p.multiProcessorCount;
    return n;
}
