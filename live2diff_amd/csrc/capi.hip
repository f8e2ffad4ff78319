// C ABI of libl2d_hip.so (see include/l2d.h): op dispatch, plan replay, hipGraph capture, timing.
// The plan executor is the native replacement of the reference's TensorRT `Engine.infer`
// (live2diff/acceleration/tensorrt/utilities.py:266-294): same role (run a static engine on a stream,
// optionally from a captured graph) -- but it works IN PLACE on the caller's KV-cache instead of copying
// every input into engine-owned buffers (utilities.py:267-268).
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "common.h"

static thread_local char g_err[512] = "";
int l2d_g_dry_run = 0;

void l2d_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int l2d_check_launch(const char *what, int tag) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        l2d_set_error("%s(tag %d): HIP launch failed: %s", what, tag, hipGetErrorString(e));
        return L2D_ELAUNCH;
    }
    return L2D_OK;
}

static int run_one(const l2d_op *op, hipStream_t s) {
    switch (op->kind) {
        case L2D_OP_IGEMM: return l2d_launch_igemm(op, s);
        case L2D_OP_GN_STATS: return l2d_launch_gn_stats(op, s);
        case L2D_OP_GN_APPLY: return l2d_launch_gn_apply(op, s);
        case L2D_OP_LAYERNORM: return l2d_launch_layernorm(op, s);
        case L2D_OP_FLASH_ATTN: return l2d_launch_flash_attn(op, s);
        case L2D_OP_TATTN_STREAM: return l2d_launch_tattn_stream(op, s);
        case L2D_OP_TATTN_WARMUP: return l2d_launch_tattn_warmup(op, s);
        case L2D_OP_SKINNY_LINEAR: return l2d_launch_skinny_linear(op, s);
        case L2D_OP_TIMESTEP_EMBED: return l2d_launch_timestep_embed(op, s);
        case L2D_OP_NCHW_TO_NHWC: return l2d_launch_nchw_to_nhwc(op, s);
        case L2D_OP_NHWC_TO_NCHW: return l2d_launch_nhwc_to_nchw(op, s);
        case L2D_OP_LCM_STEP: return l2d_launch_lcm_step(op, s);
        case L2D_OP_RING_UPDATE: return l2d_launch_ring_update(op, s);
        case L2D_OP_STREAM_SHIFT: return l2d_launch_stream_shift(op, s);
        case L2D_OP_RANDN: return l2d_launch_randn(op, s);
        case L2D_OP_RESIZE_BILINEAR: return l2d_launch_resize_bilinear(op, s);
        case L2D_OP_MINMAX: return l2d_launch_minmax(op, s);
        case L2D_OP_DEPTH_NORM_RESIZE: return l2d_launch_depth_norm_resize(op, s);
        case L2D_OP_STEM7X7: return l2d_launch_stem7x7(op, s);
        case L2D_OP_RESAMPLE_NHWC: return l2d_launch_resample_nhwc(op, s);
        case L2D_OP_EW: return l2d_launch_ew(op, s);
        case L2D_OP_ROWGEMM: return l2d_launch_rowgemm(op, s);
        case L2D_OP_PCONV: return l2d_launch_pconv(op, s);
        case L2D_OP_WSGEMM: return l2d_launch_wsgemm(op, s);
        case L2D_OP_ROWCHAIN: return l2d_launch_rowchain(op, s);
        case L2D_OP_CCONV: return l2d_launch_cconv(op, s);
        case L2D_OP_COPY: {
            if (!op->p[0] || !op->p[1] || op->l[0] <= 0) {
                l2d_set_error("copy(tag %d): invalid arguments", op->tag);
                return L2D_EINVAL;
            }
            if (l2d_g_dry_run) return L2D_OK;
            hipError_t e = hipMemcpyAsync(op->p[1], op->p[0], (size_t)op->l[0], hipMemcpyDeviceToDevice, s);
            if (e != hipSuccess) {
                l2d_set_error("copy(tag %d): %s", op->tag, hipGetErrorString(e));
                return L2D_ELAUNCH;
            }
            return L2D_OK;
        }
    }
    l2d_set_error("unknown op kind %d (tag %d)", op->kind, op->tag);
    return L2D_EINVAL;
}

extern "C" {

int l2d_abi_version(void) { return L2D_ABI_VERSION; }

int l2d_set_dry_run(int on) {
    l2d_g_dry_run = on ? 1 : 0;
    return L2D_OK;
}

const char *l2d_last_error(void) { return g_err; }

int l2d_device_check(char *name, int name_len) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
        l2d_set_error("no HIP device");
        return L2D_ENODEV;
    }
    if (name && name_len > 0) {
        snprintf(name, (size_t)name_len, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        l2d_set_error("device arch %s is not gfx950", prop.gcnArchName);
        return L2D_ENODEV;
    }
    return L2D_OK;
}

int l2d_run_ops(const l2d_op *ops, int n, void *stream) {
    if (!ops || n < 0) {
        l2d_set_error("run_ops: invalid arguments");
        return L2D_EINVAL;
    }
    hipStream_t s = (hipStream_t)stream;
    for (int i = 0; i < n; ++i) {
        int rc = run_one(&ops[i], s);
        if (rc != L2D_OK) return rc;
    }
    return L2D_OK;
}

struct l2d_graph {
    hipGraph_t graph;
    hipGraphExec_t exec;
};

int l2d_graph_create(const l2d_op *ops, int n, void *stream, void **graph_out) {
    if (!ops || n <= 0 || !graph_out) {
        l2d_set_error("graph_create: invalid arguments");
        return L2D_EINVAL;
    }
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) {
        l2d_set_error("graph_create: BeginCapture: %s", hipGetErrorString(e));
        return L2D_ELAUNCH;
    }
    int rc = l2d_run_ops(ops, n, stream);
    hipGraph_t g = nullptr;
    e = hipStreamEndCapture(s, &g);
    if (rc != L2D_OK) {
        if (g) hipGraphDestroy(g);
        return rc;
    }
    if (e != hipSuccess || !g) {
        l2d_set_error("graph_create: EndCapture: %s", hipGetErrorString(e));
        return L2D_ELAUNCH;
    }
    hipGraphExec_t ex = nullptr;
    e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
    if (e != hipSuccess) {
        hipGraphDestroy(g);
        l2d_set_error("graph_create: Instantiate: %s", hipGetErrorString(e));
        return L2D_ELAUNCH;
    }
    l2d_graph *h = new l2d_graph{g, ex};
    *graph_out = h;
    return L2D_OK;
}

int l2d_graph_launch(void *graph, void *stream) {
    if (!graph) {
        l2d_set_error("graph_launch: null graph");
        return L2D_EINVAL;
    }
    hipError_t e = hipGraphLaunch(((l2d_graph *)graph)->exec, (hipStream_t)stream);
    if (e != hipSuccess) {
        l2d_set_error("graph_launch: %s", hipGetErrorString(e));
        return L2D_ELAUNCH;
    }
    return L2D_OK;
}

int l2d_graph_destroy(void *graph) {
    if (!graph) return L2D_OK;
    l2d_graph *h = (l2d_graph *)graph;
    hipGraphExecDestroy(h->exec);
    hipGraphDestroy(h->graph);
    delete h;
    return L2D_OK;
}

int l2d_time_ops(const l2d_op *ops, int n, void *stream, int reps, float *ms_out) {
    if (!ops || n <= 0 || reps <= 0 || !ms_out) {
        l2d_set_error("time_ops: invalid arguments");
        return L2D_EINVAL;
    }
    hipStream_t s = (hipStream_t)stream;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0, s);
    int rc = L2D_OK;
    for (int r = 0; r < reps && rc == L2D_OK; ++r) rc = l2d_run_ops(ops, n, stream);
    hipEventRecord(e1, s);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    *ms_out = ms / (float)reps;
    return rc;
}

// In-frame duration of every launch without a profiler: one hipEvent in front of each op and one behind the last, `reps`
// passes; us_out[i] = mean time from op i's event to the next one (its launch incl. the gap to its successor), on `stream`.
int l2d_time_each(const l2d_op *ops, int n, void *stream, int reps, float *us_out) {
    if (!ops || n <= 0 || reps <= 0 || !us_out) {
        l2d_set_error("time_each: invalid arguments");
        return L2D_EINVAL;
    }
    hipStream_t s = (hipStream_t)stream;
    hipEvent_t *ev = new hipEvent_t[n + 1];
    for (int i = 0; i <= n; ++i) hipEventCreate(&ev[i]);
    for (int i = 0; i < n; ++i) us_out[i] = 0.f;
    int rc = L2D_OK;
    for (int r = 0; r < reps && rc == L2D_OK; ++r) {
        for (int i = 0; i < n && rc == L2D_OK; ++i) {
            hipEventRecord(ev[i], s);
            rc = run_one(&ops[i], s);
        }
        hipEventRecord(ev[n], s);
        hipEventSynchronize(ev[n]);
        for (int i = 0; i < n && rc == L2D_OK; ++i) {
            float ms = 0.f;
            hipEventElapsedTime(&ms, ev[i], ev[i + 1]);
            us_out[i] += 1000.f * ms / (float)reps;
        }
    }
    for (int i = 0; i <= n; ++i) hipEventDestroy(ev[i]);
    delete[] ev;
    return rc;
}

}  // extern "C"
