#include "common.hip.h"

extern "C" int qcn_version(void) { return 200; }

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
