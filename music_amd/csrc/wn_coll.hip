// The one collective of the path (the reference's nn.DataParallel gradient reduction, wavenet/train.py:116-122):
// a sum over the ranks of ONE flat fp32 buffer, as a thin wrapper of RCCL's ncclAllReduce on a communicator the caller owns.
// libwavenet_hip.so does not link RCCL: the five entry points it needs are resolved at first use from the process image (a host
// that has RCCL loaded - torch does - gets exactly that copy: two copies of RCCL in one process are not an option) or, failing
// that, from librccl.so on the loader path; without either every function here returns -5 and says so.
#include <dlfcn.h>
#include <stdint.h>
#include <string.h>
#include "wn_common.h"
#include "wn_kernels.h"

namespace {
// the part of rccl.h this file uses (ABI of NCCL 2.x: the enum values and the 128-byte id are fixed by it)
typedef struct { char internal[128]; } WnNcclId;
typedef int (*fn_get_id)(WnNcclId*);
typedef int (*fn_init_rank)(void**, int, WnNcclId, int);
typedef int (*fn_destroy)(void*);
typedef int (*fn_all_reduce)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef const char* (*fn_err)(int);
const int kNcclFloat32 = 7, kNcclSum = 0;

struct Rccl {
    fn_get_id get_id = nullptr;
    fn_init_rank init_rank = nullptr;
    fn_destroy destroy = nullptr;
    fn_all_reduce all_reduce = nullptr;
    fn_err err = nullptr;
    bool ok = false;
};

const Rccl& rccl() {
    static const Rccl r = [] {
        Rccl x;
        void* h = RTLD_DEFAULT;
        if (!dlsym(RTLD_DEFAULT, "ncclAllReduce")) {
            h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
            if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
            if (!h) return x;
        }
        x.get_id = (fn_get_id)dlsym(h, "ncclGetUniqueId");
        x.init_rank = (fn_init_rank)dlsym(h, "ncclCommInitRank");
        x.destroy = (fn_destroy)dlsym(h, "ncclCommDestroy");
        x.all_reduce = (fn_all_reduce)dlsym(h, "ncclAllReduce");
        x.err = (fn_err)dlsym(h, "ncclGetErrorString");
        x.ok = x.get_id && x.init_rank && x.destroy && x.all_reduce;
        return x;
    }();
    return r;
}

int fail(const Rccl& r, const char* what, int rc) {
    char msg[256];
    snprintf(msg, sizeof(msg), "%s: RCCL error %d (%s)", what, rc, r.err ? r.err(rc) : "?");
    return wn_set_error_msg(-6, msg);
}
int need(const Rccl& r) {
    return r.ok ? 0 : wn_set_error_msg(-5, "RCCL is not loaded in this process and librccl.so is not on the loader path");
}
}  // namespace

int wn_coll_loaded() { return rccl().ok ? 1 : 0; }

int wn_coll_unique_id(char* id128) {
    const Rccl& r = rccl();
    if (need(r)) return -5;
    if (!id128) return wn_set_error_msg(-4, "wn_comm_unique_id: null argument");
    WnNcclId id;
    const int rc = r.get_id(&id);
    if (rc) return fail(r, "ncclGetUniqueId", rc);
    memcpy(id128, id.internal, sizeof(id.internal));
    return 0;
}

int wn_coll_create(int nranks, int rank, const char* id128, void** comm) {
    const Rccl& r = rccl();
    if (need(r)) return -5;
    if (!id128 || !comm || nranks < 1 || rank < 0 || rank >= nranks) return wn_set_error_msg(-4, "wn_comm_create: bad argument");
    WnNcclId id;
    memcpy(id.internal, id128, sizeof(id.internal));
    *comm = nullptr;
    const int rc = r.init_rank(comm, nranks, id, rank);
    if (rc) return fail(r, "ncclCommInitRank", rc);
    return 0;
}

int wn_coll_destroy(void* comm) {
    const Rccl& r = rccl();
    if (need(r)) return -5;
    if (!comm) return 0;
    const int rc = r.destroy(comm);
    if (rc) return fail(r, "ncclCommDestroy", rc);
    return 0;
}

int wn_coll_allreduce_flat(void* comm, float* buf, int64_t n, hipStream_t st) {
    const Rccl& r = rccl();
    if (need(r)) return -5;
    if (!comm || (!buf && n > 0) || n < 0) return wn_set_error_msg(-4, "wn_allreduce_flat: bad argument");
    if (n == 0) return 0;
    const int rc = r.all_reduce(buf, buf, (size_t)n, kNcclFloat32, kNcclSum, comm, st);
    if (rc) return fail(r, "ncclAllReduce", rc);
    return 0;
}
