// C-ABI plumbing: error string, device info, hipGraph capture/replay helpers, HIP-event timing.
#include <stdarg.h>
#include "bc_common.h"

static thread_local char g_err[512] = "";

void bc_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* bc_last_error(void) { return g_err; }
extern "C" int bc_version(void) { return 100; }
extern "C" int bc_sizeof_gemm(void) { return (int)sizeof(BcGemm); }

extern "C" int bc_device_info(int* out4) {
    BC_CHECK_ARG(out4 != nullptr, "bc_device_info: null output");
    int dev = 0;
    BC_CHECK_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    BC_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
    out4[0] = prop.multiProcessorCount;
    out4[1] = prop.warpSize;
    out4[2] = (int)(prop.sharedMemPerBlock / 1024);
    out4[3] = prop.major * 100 + prop.minor;
    return 0;
}

extern "C" int bc_graph_begin(bc_stream stream) {
    BC_CHECK_HIP(hipStreamBeginCapture(reinterpret_cast<hipStream_t>(stream), hipStreamCaptureModeThreadLocal));
    return 0;
}

extern "C" int bc_graph_end(bc_stream stream, void** graph_exec_out) {
    BC_CHECK_ARG(graph_exec_out != nullptr, "bc_graph_end: null output");
    hipGraph_t graph = nullptr;
    BC_CHECK_HIP(hipStreamEndCapture(reinterpret_cast<hipStream_t>(stream), &graph));
    hipGraphExec_t exec = nullptr;
    hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) {
        bc_set_error("hipGraphInstantiate failed: %s", hipGetErrorString(e));
        return 2;
    }
    *graph_exec_out = exec;
    return 0;
}

extern "C" int bc_graph_launch(void* graph_exec, bc_stream stream) {
    BC_CHECK_ARG(graph_exec != nullptr, "bc_graph_launch: null graph");
    BC_CHECK_HIP(hipGraphLaunch(reinterpret_cast<hipGraphExec_t>(graph_exec), reinterpret_cast<hipStream_t>(stream)));
    return 0;
}

extern "C" int bc_graph_destroy(void* graph_exec) {
    if (graph_exec) BC_CHECK_HIP(hipGraphExecDestroy(reinterpret_cast<hipGraphExec_t>(graph_exec)));
    return 0;
}

extern "C" int bc_event_create(void** ev) {
    BC_CHECK_ARG(ev != nullptr, "bc_event_create: null output");
    hipEvent_t e;
    BC_CHECK_HIP(hipEventCreate(&e));
    *ev = e;
    return 0;
}

extern "C" int bc_event_create_sync(void** ev) {
    BC_CHECK_ARG(ev != nullptr, "bc_event_create_sync: null output");
    hipEvent_t e;
    BC_CHECK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    *ev = e;
    return 0;
}

extern "C" int bc_stream_wait_event(bc_stream stream, void* ev) {
    BC_CHECK_HIP(hipStreamWaitEvent(reinterpret_cast<hipStream_t>(stream), reinterpret_cast<hipEvent_t>(ev), 0));
    return 0;
}

extern "C" int bc_event_record(void* ev, bc_stream stream) {
    BC_CHECK_HIP(hipEventRecord(reinterpret_cast<hipEvent_t>(ev), reinterpret_cast<hipStream_t>(stream)));
    return 0;
}

extern "C" int bc_event_elapsed_ms(void* start, void* stop, float* ms) {
    BC_CHECK_ARG(ms != nullptr, "bc_event_elapsed_ms: null output");
    BC_CHECK_HIP(hipEventSynchronize(reinterpret_cast<hipEvent_t>(stop)));
    BC_CHECK_HIP(hipEventElapsedTime(ms, reinterpret_cast<hipEvent_t>(start), reinterpret_cast<hipEvent_t>(stop)));
    return 0;
}

extern "C" int bc_event_destroy(void* ev) {
    if (ev) BC_CHECK_HIP(hipEventDestroy(reinterpret_cast<hipEvent_t>(ev)));
    return 0;
}
