#include <mutex>
#include <vector>

#include "common.hip.h"

namespace {
struct ProfRecord { hipEvent_t start, stop; double units; int kind; };
std::mutex g_prof_mutex;
std::vector<ProfRecord *> g_prof_log;
unsigned g_prof_mask = 0;   // bit k: record launches of kernel family k
}  // namespace

void *qcn_prof_begin(int kind, double alg_units, hipStream_t stream)
{
    {
        std::lock_guard<std::mutex> lock(g_prof_mutex);
        if (!((g_prof_mask >> kind) & 1u)) return nullptr;
    }
    ProfRecord *rec = new ProfRecord();
    rec->units = alg_units;
    rec->kind = kind;
    hipEventCreate(&rec->start);
    hipEventCreate(&rec->stop);
    hipEventRecord(rec->start, stream);
    return rec;
}

void qcn_prof_end(void *token, hipStream_t stream)
{
    if (!token) return;
    ProfRecord *rec = static_cast<ProfRecord *>(token);
    hipEventRecord(rec->stop, stream);
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    g_prof_log.push_back(rec);
}

extern "C" int qcn_prof_enable(int on)
{
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    g_prof_mask = on < 0 ? ~0u : (unsigned)on;
    return QCN_OK;
}

extern "C" int qcn_prof_collect(int kind, int64_t *launches_host, double *millis_host, double *alg_units_host)
{
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    int64_t n = 0;
    double ms = 0.0, units = 0.0;
    std::vector<ProfRecord *> rest;
    for (ProfRecord *r : g_prof_log) {
        if (kind >= 0 && r->kind != kind) { rest.push_back(r); continue; }
        float t = 0.f;
        if (hipEventSynchronize(r->stop) == hipSuccess && hipEventElapsedTime(&t, r->start, r->stop) == hipSuccess) {
            ++n; ms += t; units += r->units;
        }
        hipEventDestroy(r->start);
        hipEventDestroy(r->stop);
        delete r;
    }
    g_prof_log.swap(rest);
    if (launches_host) *launches_host = n;
    if (millis_host) *millis_host = ms;
    if (alg_units_host) *alg_units_host = units;
    return QCN_OK;
}

namespace { int g_matmul_mode = QCN_MATMUL_SPLIT_F16; }
int qcn_matmul_mode() { return g_matmul_mode; }

extern "C" int qcn_set_matmul_mode(int mode)
{
    if (mode != QCN_MATMUL_EXACT_F32 && mode != QCN_MATMUL_SPLIT_F16) return QCN_ERR_INVALID_ARGUMENT;
    g_matmul_mode = mode;
    return QCN_OK;
}

extern "C" int qcn_get_matmul_mode(void) { return g_matmul_mode; }

extern "C" int qcn_version(void) { return 100; }

extern "C" const char *qcn_status_string(int status)
{
    switch (status) {
    case QCN_OK: return "ok";
    case QCN_ERR_INVALID_ARGUMENT: return "invalid argument";
    case QCN_ERR_UNSUPPORTED: return "unsupported shape or size";
    case QCN_ERR_WORKSPACE_TOO_SMALL: return "workspace too small";
    case QCN_ERR_LAUNCH_FAILED: return "HIP launch failed";
    case QCN_ERR_TOO_LARGE: return "size exceeds the fixed capacity of this allocation-free entry point";
    default: return "unknown status";
    }
}
