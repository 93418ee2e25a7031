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

namespace dir {
// launch log of the calling thread: the last LOG_MAX kernel names since dir_launch_log_reset()
constexpr int LOG_MAX = 32;
static thread_local const char* g_log[LOG_MAX];
static thread_local unsigned g_nlog = 0;            // launches since the last reset, SATURATING at NLOG_SAT: a serving process makes
constexpr unsigned NLOG_SAT = 0x7fffffffu;         // thousands of launches per step and never resets (only bench.py's profiling does)
void note_kernel(const char* name) {
    if (g_nlog < (unsigned)LOG_MAX) g_log[g_nlog] = name;
    if (g_nlog < NLOG_SAT) ++g_nlog;
}
}  // namespace dir

extern "C" void dir_launch_log_reset(void) { dir::g_nlog = 0; }

// as if `times` launches of kernel `name` had been noted (O(LOG_MAX), not O(times)): lets the CPU tests drive the counter past
// LOG_MAX and up to its saturation point without a GPU
extern "C" void dir_launch_log_note(const char* name, long long times) {
    static thread_local char keep[64];
    if (!name || times <= 0) return;
    strncpy(keep, name, sizeof(keep) - 1);
    keep[sizeof(keep) - 1] = 0;
    while (times > 0 && dir::g_nlog < (unsigned)dir::LOG_MAX) { dir::note_kernel(keep); --times; }
    const unsigned long long room = dir::NLOG_SAT - dir::g_nlog;
    dir::g_nlog += (unsigned)((unsigned long long)times < room ? (unsigned long long)times : room);
}

extern "C" int dir_launch_log_get(char* buf_host, int len) {
    // names as written at the launch site, template arguments dropped: "conv_pipe_kernel,pgcn_layer_kernel,..."
    int pos = 0;
    const int n = dir::g_nlog < (unsigned)dir::LOG_MAX ? (int)dir::g_nlog : dir::LOG_MAX;
    for (int i = 0; i < n && buf_host && len > 0; ++i) {
        const char* s = dir::g_log[i];
        while (*s == '(' || *s == ' ') ++s;
        if (i && pos < len - 1) buf_host[pos++] = ',';
        for (; *s && *s != '<' && *s != ')' && *s != ' ' && pos < len - 1; ++s) buf_host[pos++] = *s;
    }
    if (buf_host && len > 0) buf_host[pos] = 0;
    return (int)dir::g_nlog;          // saturates at INT_MAX
}

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
