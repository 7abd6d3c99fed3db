// libvocr: the one exchange step of the data-parallel path — a SUM all-reduce of the flat fp32 gradient across the GPUs of
// a node over xGMI — for hosts that do not go through torch.distributed.  Thin wrapper over RCCL, bound lazily with
// dlopen so that libvocr.so has no link-time dependency on it (single-GPU users never load it) and so that a process
// which already carries an RCCL (PyTorch-ROCm does) keeps using that one copy.
// Reference side: nn.DataParallel's gradient reduction, src/models/cnnlstm.py:198-199, src/train_cnn_lstm.py:333.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include "../../include/vocr.h"

void vocr_set_error(const char* fmt, ...);

namespace {

// Minimal RCCL ABI (rccl.h): opaque communicator, 128-byte unique id passed by value, enums as ints.
struct UniqueId { char internal[128]; };
typedef void* Comm;
typedef int (*GetUniqueIdFn)(UniqueId*);
typedef int (*CommInitRankFn)(Comm*, int, UniqueId, int);
typedef int (*CommDestroyFn)(Comm);
typedef int (*AllReduceFn)(const void*, void*, size_t, int, int, Comm, hipStream_t);
typedef const char* (*GetErrorStringFn)(int);
constexpr int kFloat32 = 7;   // ncclFloat32
constexpr int kSum = 0;       // ncclSum

struct Api {
    void* lib = nullptr;
    GetUniqueIdFn get_unique_id = nullptr;
    CommInitRankFn comm_init_rank = nullptr;
    CommDestroyFn comm_destroy = nullptr;
    AllReduceFn all_reduce = nullptr;
    GetErrorStringFn error_string = nullptr;
};

Api* api() {
    static Api a;
    static bool tried = false;
    if (!tried) {
        tried = true;
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names)                                   // a copy the process already holds, first
            if ((a.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
        if (!a.lib)
            for (const char* n : names)
                if ((a.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
        if (a.lib) {
            a.get_unique_id = (GetUniqueIdFn)dlsym(a.lib, "ncclGetUniqueId");
            a.comm_init_rank = (CommInitRankFn)dlsym(a.lib, "ncclCommInitRank");
            a.comm_destroy = (CommDestroyFn)dlsym(a.lib, "ncclCommDestroy");
            a.all_reduce = (AllReduceFn)dlsym(a.lib, "ncclAllReduce");
            a.error_string = (GetErrorStringFn)dlsym(a.lib, "ncclGetErrorString");
        }
    }
    if (!a.lib || !a.get_unique_id || !a.comm_init_rank || !a.comm_destroy || !a.all_reduce) return nullptr;
    return &a;
}

int fail(const char* what, int rc) {
    Api* a = api();
    vocr_set_error("%s: RCCL error %d (%s)", what, rc, (a && a->error_string) ? a->error_string(rc) : "?");
    return VOCR_ECOMM;
}

}  // namespace

struct vocr_comm {
    Comm comm;
    int nranks, rank, device;
};

extern "C" int vocr_comm_unique_id(void* id128) {
    if (!id128) { vocr_set_error("vocr_comm_unique_id: null pointer"); return VOCR_EINVAL; }
    Api* a = api();
    if (!a) { vocr_set_error("vocr_comm_unique_id: librccl.so not found"); return VOCR_ECOMM; }
    UniqueId id;
    const int rc = a->get_unique_id(&id);
    if (rc != 0) return fail("vocr_comm_unique_id", rc);
    memcpy(id128, &id, sizeof(id));
    return VOCR_OK;
}

extern "C" int vocr_comm_create(vocr_comm** out, const void* id128, int nranks, int rank, int device) {
    if (!out || !id128 || nranks < 1 || rank < 0 || rank >= nranks || device < 0) {
        vocr_set_error("vocr_comm_create: bad argument (nranks=%d rank=%d device=%d)", nranks, rank, device);
        return VOCR_EINVAL;
    }
    Api* a = api();
    if (!a) { vocr_set_error("vocr_comm_create: librccl.so not found"); return VOCR_ECOMM; }
    if (hipSetDevice(device) != hipSuccess) { vocr_set_error("vocr_comm_create: hipSetDevice(%d) failed", device); return VOCR_ENODEVICE; }
    UniqueId id;
    memcpy(&id, id128, sizeof(id));
    Comm c = nullptr;
    const int rc = a->comm_init_rank(&c, nranks, id, rank);
    if (rc != 0) return fail("vocr_comm_create", rc);
    vocr_comm* h = new vocr_comm{c, nranks, rank, device};
    *out = h;
    return VOCR_OK;
}

extern "C" int vocr_allreduce_sum_f32(vocr_comm* comm, float* buf, size_t count, void* stream) {
    if (!comm || !buf) { vocr_set_error("vocr_allreduce_sum_f32: null pointer"); return VOCR_EINVAL; }
    if (count == 0) return VOCR_OK;
    Api* a = api();
    if (!a) { vocr_set_error("vocr_allreduce_sum_f32: librccl.so not found"); return VOCR_ECOMM; }
    const int rc = a->all_reduce(buf, buf, count, kFloat32, kSum, comm->comm, (hipStream_t)stream);
    if (rc != 0) return fail("vocr_allreduce_sum_f32", rc);
    return VOCR_OK;
}

extern "C" int vocr_comm_destroy(vocr_comm* comm) {
    if (!comm) return VOCR_OK;
    Api* a = api();
    int rc = 0;
    if (a && comm->comm) rc = a->comm_destroy(comm->comm);
    delete comm;
    if (rc != 0) return fail("vocr_comm_destroy", rc);
    return VOCR_OK;
}
