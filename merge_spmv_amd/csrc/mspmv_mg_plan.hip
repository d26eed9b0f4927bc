// mspmv_mg_plan.hip -- the multi-GPU merge-partitioned CsrMV operator below the C ABI
// (include/mspmv.h: mspmv_mg_plan_* / mspmv_mg_csrmv / mspmv_mg_allgather_rows).
//
// New design -- the reference is single-device; what it offers is the claim that the merge
// decomposition "is suitable for recursively partitioning CSR datasets" (README.md:5) and the
// thread-level scheme of cpu_spmv.cpp:305-352, lifted here one level: a GPU plays the role of an
// OpenMP thread.  mspmv_mg_partition (mspmv_mg.cpp) cuts the global merge path at G equally spaced
// diagonals; part g is an ordinary local CSR matrix with one extra LAST row -- the row cut by its right
// boundary -- so the unchanged single-GPU CsrMV (csrmv_call, mspmv_api.hip) computes the part and its
// last y entry IS the carry for global row row_split[g+1].  One exchange step per SpMV moves those G
// scalars; each part then adds to its first row the carries of earlier parts whose key equals it, in
// part order (deterministic) -- cpu_spmv.cpp:348-352 lifted to parts.
//
// A plan holds the parts one PROCESS drives:
//   * all G of them (single-process form; parts may share a device, which is how a 1-GPU box runs it), or
//   * one (one process per GPU under torch.distributed.run, what bench.py --gpus N does).
// Exchange backends:
//   PEER  single-process only.  Every part's stream records an event after its fix-up; the stream of a
//         part that has carries to take waits on its sources' events (cross-device hipStreamWaitEvent) and
//         ONE small kernel reads the sources' last-row values straight out of the peers' memory over xGMI
//         and adds them.  No collective, no staging buffer, no host involvement.
//   RCCL  ncclAllGather of one element per part (sendbuff = the address of the part's last y entry),
//         inside ncclGroupStart/End when the process holds several parts, then the same add kernel
//         reading the gathered array.  librccl is dlopen'ed on first use, so the single-GPU library has
//         no link-time dependency on it (inside a torch process that is torch's own librccl.so.1).
// mspmv_mg_allgather_rows turns the row-sharded y into the replicated x of the next
// SpMV: PEER = every part pushes its owned rows straight into every replica of x (direct writes over the
// fully connected xGMI, unpadded, no host loop); RCCL = G grouped ncclBroadcast calls (the all-gather-v
// idiom), again unpadded and written in place.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <vector>

#include "../../include/mspmv.h"
#include "mspmv_internal.hpp"

namespace {

using namespace mspmv;

// ---- the few RCCL entry points used, resolved at run time ------------------------------------
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
enum { kNcclFloat = 7, kNcclDouble = 8 };          // ncclFloat32 / ncclFloat64 of rccl.h
struct Rccl {
    void *handle = nullptr;
    int (*GetUniqueId)(ncclUniqueId *) = nullptr;
    int (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    int (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Broadcast)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool ok = false;
};
Rccl &rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.handle) break;
        }
        if (!r.handle) return;
#define MSPMV_SYM(field, sym) r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.handle, sym))
        MSPMV_SYM(GetUniqueId, "ncclGetUniqueId"); MSPMV_SYM(CommInitRank, "ncclCommInitRank");
        MSPMV_SYM(CommInitAll, "ncclCommInitAll"); MSPMV_SYM(CommDestroy, "ncclCommDestroy");
        MSPMV_SYM(GroupStart, "ncclGroupStart"); MSPMV_SYM(GroupEnd, "ncclGroupEnd");
        MSPMV_SYM(AllGather, "ncclAllGather"); MSPMV_SYM(Broadcast, "ncclBroadcast");
        MSPMV_SYM(GetErrorString, "ncclGetErrorString");
#undef MSPMV_SYM
        r.ok = r.GetUniqueId && r.CommInitRank && r.CommInitAll && r.CommDestroy && r.GroupStart && r.GroupEnd &&
               r.AllGather && r.Broadcast;
    });
    return r;
}

constexpr int kErrInvalid = hipErrorInvalidValue;
constexpr int kErrRccl = hipErrorUnknown;          // an RCCL call failed (message on stderr)
constexpr int kErrNoRccl = hipErrorNotSupported;   // librccl could not be loaded

#define MG_HIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return (int) e_; } while (0)
#define MG_NCCL(expr)                                                                                   \
    do {                                                                                                \
        int r_ = (expr);                                                                                \
        if (r_ != 0) {                                                                                  \
            fprintf(stderr, "mspmv_mg: RCCL error %d (%s) at %s:%d\n", r_,                              \
                    rccl().GetErrorString ? rccl().GetErrorString(r_) : "?", __FILE__, __LINE__);       \
            return kErrRccl;                                                                            \
        }                                                                                               \
    } while (0)

// y_first[0] += sum over the sources, in part order.  PEER: src[k] points at the last y entry of an
// earlier part (possibly in another GPU's memory); RCCL: at the gathered carry array's slots.
template <typename V>
__global__ void mg_take_carries_kernel(V *__restrict__ y_first, const V *const *__restrict__ src, int n)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    V acc = *y_first;
    for (int k = 0; k < n; ++k) acc += *src[k];
    *y_first = acc;
}

// Push `count` owned rows into every replica of x (dst[d] already offset to this part's row range).  blockIdx.y is
// the replica: the writes to the different peers are in flight together, one xGMI link each (the links are
// point-to-point, so a part's push to seven peers uses seven links at once), while y is read from local HBM / L2.
template <typename V>
__global__ __launch_bounds__(256) void mg_push_rows_kernel(const V *__restrict__ y, V *const *__restrict__ dst, long long count)
{
    const long long stride = (long long) gridDim.x * blockDim.x;
    V *__restrict__ out = dst[blockIdx.y];
    for (long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) out[i] = y[i];
}

struct Part {
    int id = 0, device = 0, replica = 0;
    long long row_begin = 0, row_end = 0, nz_begin = 0, nz_end = 0;
    int local_rows = 0, local_nnz = 0;
    long long owned = 0;
    hipStream_t stream = nullptr;
    void *temp = nullptr; size_t temp_bytes = 0;
    const void *values = nullptr; const int32_t *offsets = nullptr; const int32_t *cols = nullptr;
    bool attached = false;
    void *y = nullptr;                    // local_rows entries, plan-owned
    void *carries = nullptr;              // RCCL: gathered carries, G entries
    const void **src_table = nullptr;     // device array of the addresses to add to y[0]
    int nsrc = 0;
    std::vector<int> sources;             // part ids whose carry lands on this part's first row
    void **push_table = nullptr;          // device array: this part's row range inside every replica of x
    hipEvent_t done = nullptr, applied = nullptr, pushed = nullptr;
    bool applied_valid = false, pushed_valid = false, done_valid = false;
    ncclComm_t comm = nullptr;
    void *hot = nullptr; size_t hot_bytes = 0;      // hot-column plan of the part (mspmv_mg_plan_hot_columns): its own temp, indices, x
};

struct Replica { int device = 0; void *x = nullptr; };

}  // namespace

struct mspmv_mg_plan {
    int parts = 0, value_bytes = 0, exchange = 0;
    long long rows = 0, cols = 0;
    bool whole = false;                   // this process holds every part
    std::vector<long long> row_split, nz_split;
    std::vector<Part> local;
    std::vector<Replica> replicas;
    unsigned long long steps = 0;
};

namespace {

int part_index(const mspmv_mg_plan *p, int id)
{
    for (size_t i = 0; i < p->local.size(); ++i) if (p->local[i].id == id) return (int) i;
    return -1;
}

template <typename V>
int run_spmv(mspmv_mg_plan *plan)
{
    CallExtra ex; ex.phase = PHASE_SKIP_COORDS;
    // 1. every part's local SpMV (its last local row is the part's carry)
    for (Part &q : plan->local) {
        if (!q.attached) return kErrInvalid;
        MG_HIP(hipSetDevice(q.device));
        // x must be complete (a preceding all-gather pushed into this device's replica from every part) and
        // nobody may still be reading what this SpMV overwrites (the carry readers of the previous step)
        for (Part &o : plan->local) {
            if (o.pushed_valid) MG_HIP(hipStreamWaitEvent(q.stream, o.pushed, 0));
            if (&o != &q && o.applied_valid && std::find(o.sources.begin(), o.sources.end(), q.id) != o.sources.end())
                MG_HIP(hipStreamWaitEvent(q.stream, o.applied, 0));
        }
        size_t tb = q.temp_bytes;
        int st;
        if (q.hot) {
            // the part's columns renumbered by reference count (mspmv_hotcols.hip): x is permuted into the part's numbering first
            if (sizeof(V) == 4)
                st = mspmv_csrmv_hotcols_apply_f32(q.hot, q.hot_bytes, static_cast<const float *>(q.values), q.offsets,
                                                   static_cast<const float *>(plan->replicas[q.replica].x), static_cast<float *>(q.y), q.local_rows,
                                                   (int32_t) plan->cols, q.local_nnz, 1.f, 0.f, q.stream, 0);
            else
                st = mspmv_csrmv_hotcols_apply_f64(q.hot, q.hot_bytes, static_cast<const double *>(q.values), q.offsets,
                                                   static_cast<const double *>(plan->replicas[q.replica].x), static_cast<double *>(q.y), q.local_rows,
                                                   (int32_t) plan->cols, q.local_nnz, 1.0, 0.0, q.stream, 0);
        } else
            st = csrmv_call<V>(q.temp, &tb, static_cast<const V *>(q.values), q.offsets, q.cols,
                               static_cast<const V *>(plan->replicas[q.replica].x), static_cast<V *>(q.y), q.local_rows,
                               (int32_t) plan->cols, q.local_nnz, (V) 1, (V) 0, false, q.stream, 0, ex);
        if (st != 0) return st;
        MG_HIP(hipEventRecord(q.done, q.stream)); q.done_valid = true;
    }
    if (plan->parts == 1 && plan->exchange != MSPMV_MG_EXCHANGE_RCCL) { ++plan->steps; return 0; }
    // 2. the one exchange + the owners' adds (a one-part RCCL plan still issues its all-gather: that is how a
    //    single-GPU box exercises the RCCL path end to end)
    if (plan->exchange == MSPMV_MG_EXCHANGE_RCCL) {
        Rccl &r = rccl();
        const bool group = plan->local.size() > 1;
        if (group) MG_NCCL(r.GroupStart());
        for (Part &q : plan->local) {
            MG_HIP(hipSetDevice(q.device));
            MG_NCCL(r.AllGather(static_cast<const V *>(q.y) + (q.local_rows - 1), q.carries, 1,
                                sizeof(V) == 4 ? kNcclFloat : kNcclDouble, q.comm, q.stream));
        }
        if (group) MG_NCCL(r.GroupEnd());
    }
    for (Part &q : plan->local) {
        if (q.nsrc == 0) continue;
        MG_HIP(hipSetDevice(q.device));
        if (plan->exchange == MSPMV_MG_EXCHANGE_PEER)
            for (int src : q.sources) MG_HIP(hipStreamWaitEvent(q.stream, plan->local[part_index(plan, src)].done, 0));
        hipLaunchKernelGGL((mg_take_carries_kernel<V>), dim3(1), dim3(64), 0, q.stream, static_cast<V *>(q.y),
                           reinterpret_cast<const V *const *>(q.src_table), q.nsrc);
        MG_HIP(hipGetLastError());
        MG_HIP(hipEventRecord(q.applied, q.stream)); q.applied_valid = true;
    }
    ++plan->steps;
    return 0;
}

template <typename V>
int run_allgather(mspmv_mg_plan *plan)
{
    if (plan->rows != plan->cols) return kErrInvalid;          // y -> x needs a square operator
    if (plan->exchange == MSPMV_MG_EXCHANGE_RCCL) {
        Rccl &r = rccl();
        MG_NCCL(r.GroupStart());
        for (int g = 0; g < plan->parts; ++g) {
            const long long count = plan->row_split[g + 1] - plan->row_split[g];
            if (count == 0) continue;
            for (Part &q : plan->local) {
                MG_HIP(hipSetDevice(q.device));
                V *recv = static_cast<V *>(plan->replicas[q.replica].x) + plan->row_split[g];
                MG_NCCL(r.Broadcast(q.id == g ? q.y : (const void *) recv, recv, (size_t) count,
                                    sizeof(V) == 4 ? kNcclFloat : kNcclDouble, g, q.comm, q.stream));
            }
        }
        MG_NCCL(r.GroupEnd());
        return 0;
    }
    // PEER: nobody may still be reading x (every part's SpMV done) before the pushes overwrite it
    for (Part &q : plan->local) {
        MG_HIP(hipSetDevice(q.device));
        for (Part &o : plan->local) if (&o != &q && o.done_valid) MG_HIP(hipStreamWaitEvent(q.stream, o.done, 0));
        if (q.owned > 0) {
            const unsigned grid = (unsigned) std::min<long long>((q.owned + 255) / 256, 1024);
            hipLaunchKernelGGL((mg_push_rows_kernel<V>), dim3(grid, (unsigned) plan->replicas.size()), dim3(256), 0, q.stream,
                               static_cast<const V *>(q.y), reinterpret_cast<V *const *>(q.push_table), q.owned);
            MG_HIP(hipGetLastError());
        }
        MG_HIP(hipEventRecord(q.pushed, q.stream)); q.pushed_valid = true;
    }
    return 0;
}

int destroy(mspmv_mg_plan *plan)
{
    if (!plan) return 0;
    for (Part &q : plan->local) {
        (void) hipSetDevice(q.device);
        if (q.stream) (void) hipStreamSynchronize(q.stream);
    }
    for (Part &q : plan->local) {
        (void) hipSetDevice(q.device);
        if (q.comm && rccl().ok) (void) rccl().CommDestroy(q.comm);
        if (q.done) (void) hipEventDestroy(q.done);
        if (q.applied) (void) hipEventDestroy(q.applied);
        if (q.pushed) (void) hipEventDestroy(q.pushed);
        (void) hipFree(q.temp); (void) hipFree(q.y); (void) hipFree(q.carries); (void) hipFree(q.src_table); (void) hipFree(q.push_table);
        (void) hipFree(q.hot);
        if (q.stream) (void) hipStreamDestroy(q.stream);
    }
    for (Replica &r : plan->replicas) { (void) hipSetDevice(r.device); (void) hipFree(r.x); }
    delete plan;
    return 0;
}

}  // namespace

extern "C" {

int mspmv_mg_unique_id(void *id128)
{
    if (!id128) return kErrInvalid;
    if (!rccl().ok) return kErrNoRccl;
    ncclUniqueId id;
    MG_NCCL(rccl().GetUniqueId(&id));
    memcpy(id128, &id, sizeof(id));
    return 0;
}

int mspmv_mg_plan_create(mspmv_mg_plan_t **out, int32_t parts, int32_t local_parts, const int32_t *part_ids,
                         const int32_t *device_ids, const int64_t *row_split, const int64_t *nz_split, int64_t cols,
                         int32_t value_bytes, int32_t exchange, const void *id128)
{
    if (!out || parts < 1 || parts > MSPMV_MG_MAX_PARTS || local_parts < 1 || local_parts > parts || !part_ids || !device_ids ||
        !row_split || !nz_split || cols < 0 || cols > 0x7fffffffLL || (value_bytes != 4 && value_bytes != 8) || exchange < 0 ||
        exchange > MSPMV_MG_EXCHANGE_PEER)
        return kErrInvalid;
    *out = nullptr;
    int ndev = 0, prev_dev = 0;
    MG_HIP(hipGetDeviceCount(&ndev));
    MG_HIP(hipGetDevice(&prev_dev));
    mspmv_mg_plan *plan = new mspmv_mg_plan;
    plan->parts = parts; plan->value_bytes = value_bytes; plan->cols = cols; plan->rows = row_split[parts];
    plan->whole = local_parts == parts && !id128;     // (an id makes even a 1-rank job take the multi-process path)
    plan->row_split.assign(row_split, row_split + parts + 1);
    plan->nz_split.assign(nz_split, nz_split + parts + 1);
    auto fail = [&](int code) { destroy(plan); (void) hipSetDevice(prev_dev); return code; };
    std::vector<bool> seen((size_t) parts, false);
    bool distinct_devices = true;
    for (int i = 0; i < local_parts; ++i) {
        const int id = part_ids[i], dev = device_ids[i];
        if (id < 0 || id >= parts || seen[id] || dev < 0 || dev >= ndev) return fail(kErrInvalid);
        seen[id] = true;
        for (int j = 0; j < i; ++j) if (device_ids[j] == dev) distinct_devices = false;
        if (row_split[id + 1] < row_split[id] || nz_split[id + 1] < nz_split[id]) return fail(kErrInvalid);
        Part q; q.id = id; q.device = dev;
        q.row_begin = row_split[id]; q.row_end = row_split[id + 1]; q.nz_begin = nz_split[id]; q.nz_end = nz_split[id + 1];
        q.owned = q.row_end - q.row_begin;
        const long long lr = q.owned + 1, ln = q.nz_end - q.nz_begin;
        if (lr + ln > MAX_ITEMS) return fail(kErrInvalid);             // a part must fit the int32 single-GPU call
        q.local_rows = (int) lr; q.local_nnz = (int) ln;
        // carries that land on this part's first row: earlier parts whose open row is that row
        if (q.owned > 0)
            for (int j = 0; j < id; ++j) if (row_split[j + 1] == row_split[id]) q.sources.push_back(j);
        q.nsrc = (int) q.sources.size();
        plan->local.push_back(q);
    }
    // exchange backend
    if (exchange == MSPMV_MG_EXCHANGE_AUTO) {
        exchange = plan->whole ? MSPMV_MG_EXCHANGE_PEER : MSPMV_MG_EXCHANGE_RCCL;
        // peer reads need every pair of distinct devices to map each other's memory (true on an xGMI node); else RCCL
        if (exchange == MSPMV_MG_EXCHANGE_PEER && distinct_devices)
            for (int i = 0; i < local_parts && exchange == MSPMV_MG_EXCHANGE_PEER; ++i)
                for (int j = 0; j < local_parts; ++j) {
                    int can = 1;
                    if (device_ids[i] != device_ids[j] && (hipDeviceCanAccessPeer(&can, device_ids[i], device_ids[j]) != hipSuccess || !can)) {
                        (void) hipGetLastError(); exchange = MSPMV_MG_EXCHANGE_RCCL; break;
                    }
                }
    }
    if (exchange == MSPMV_MG_EXCHANGE_PEER && local_parts != parts) return fail(kErrInvalid);       // peers live in this process
    if (exchange == MSPMV_MG_EXCHANGE_RCCL && !distinct_devices) return fail(kErrInvalid);  // one RCCL rank per device
    if (exchange == MSPMV_MG_EXCHANGE_RCCL && !plan->whole && !id128) return fail(kErrInvalid);
    if (exchange == MSPMV_MG_EXCHANGE_RCCL && !rccl().ok) return fail(kErrNoRccl);
    plan->exchange = exchange;
    // replicas of x: one per distinct device
    for (Part &q : plan->local) {
        int r = -1;
        for (size_t k = 0; k < plan->replicas.size(); ++k) if (plan->replicas[k].device == q.device) r = (int) k;
        if (r < 0) { Replica rep; rep.device = q.device; plan->replicas.push_back(rep); r = (int) plan->replicas.size() - 1; }
        q.replica = r;
    }
    const size_t vb = (size_t) value_bytes;
    for (Replica &rep : plan->replicas) {
        if (hipSetDevice(rep.device) != hipSuccess) return fail(kErrInvalid);
        if (hipMalloc(&rep.x, std::max<size_t>((size_t) cols * vb, 256)) != hipSuccess) return fail(hipErrorOutOfMemory);
        if (exchange == MSPMV_MG_EXCHANGE_PEER)
            for (Replica &other : plan->replicas)
                if (other.device != rep.device) {
                    const hipError_t e = hipDeviceEnablePeerAccess(other.device, 0);
                    if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { (void) hipGetLastError(); return fail((int) e); }
                    (void) hipGetLastError();
                }
    }
    // per-part resources
    for (Part &q : plan->local) {
        if (hipSetDevice(q.device) != hipSuccess) return fail(kErrInvalid);
        if (hipStreamCreateWithFlags(&q.stream, hipStreamNonBlocking) != hipSuccess) return fail(hipErrorOutOfMemory);
        q.temp_bytes = (size_t) csrmv_temp_bytes(q.local_rows, q.local_nnz, value_bytes);
        if (hipMalloc(&q.temp, q.temp_bytes) != hipSuccess) return fail(hipErrorOutOfMemory);
        if (hipMalloc(&q.y, std::max<size_t>((size_t) q.local_rows * vb, 256)) != hipSuccess) return fail(hipErrorOutOfMemory);
        if (hipMemsetAsync(q.y, 0, (size_t) q.local_rows * vb, q.stream) != hipSuccess) return fail(kErrInvalid);
        if (hipEventCreateWithFlags(&q.done, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&q.applied, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&q.pushed, hipEventDisableTiming) != hipSuccess)
            return fail(hipErrorOutOfMemory);
        if (exchange == MSPMV_MG_EXCHANGE_RCCL) {
            if (hipMalloc(&q.carries, std::max<size_t>((size_t) parts * vb, 256)) != hipSuccess) return fail(hipErrorOutOfMemory);
            if (hipMemsetAsync(q.carries, 0, (size_t) parts * vb, q.stream) != hipSuccess) return fail(kErrInvalid);
        }
    }
    // address tables (all buffers they point into are plan-owned, so they are fixed for the plan's life)
    for (Part &q : plan->local) {
        if (hipSetDevice(q.device) != hipSuccess) return fail(kErrInvalid);
        if (q.nsrc > 0) {
            std::vector<const void *> src;
            for (int j : q.sources) {
                if (exchange == MSPMV_MG_EXCHANGE_PEER) {
                    const Part &s = plan->local[part_index(plan, j)];
                    src.push_back(static_cast<const char *>(s.y) + (size_t) (s.local_rows - 1) * vb);
                } else {
                    src.push_back(static_cast<const char *>(q.carries) + (size_t) j * vb);
                }
            }
            if (hipMalloc(reinterpret_cast<void **>(&q.src_table), src.size() * sizeof(void *)) != hipSuccess) return fail(hipErrorOutOfMemory);
            if (hipMemcpy(q.src_table, src.data(), src.size() * sizeof(void *), hipMemcpyHostToDevice) != hipSuccess) return fail(kErrInvalid);
        }
        if (exchange == MSPMV_MG_EXCHANGE_PEER) {
            std::vector<void *> dst;
            for (Replica &rep : plan->replicas) dst.push_back(static_cast<char *>(rep.x) + (size_t) q.row_begin * vb);
            if (hipMalloc(reinterpret_cast<void **>(&q.push_table), dst.size() * sizeof(void *)) != hipSuccess) return fail(hipErrorOutOfMemory);
            if (hipMemcpy(q.push_table, dst.data(), dst.size() * sizeof(void *), hipMemcpyHostToDevice) != hipSuccess) return fail(kErrInvalid);
        }
    }
    // communicators: rank == part id
    if (exchange == MSPMV_MG_EXCHANGE_RCCL) {
        Rccl &r = rccl();
        if (plan->whole) {
            std::vector<ncclComm_t> comms((size_t) parts);
            std::vector<int> devs((size_t) parts);
            for (Part &q : plan->local) devs[q.id] = q.device;
            const int st = r.CommInitAll(comms.data(), parts, devs.data());
            if (st != 0) { fprintf(stderr, "mspmv_mg: ncclCommInitAll failed (%d)\n", st); return fail(kErrRccl); }
            for (Part &q : plan->local) q.comm = comms[q.id];
        } else {
            ncclUniqueId id; memcpy(&id, id128, sizeof(id));
            const bool group = plan->local.size() > 1;
            if (group) (void) r.GroupStart();
            int st = 0;
            for (Part &q : plan->local) {
                if (hipSetDevice(q.device) != hipSuccess) return fail(kErrInvalid);
                const int s = r.CommInitRank(&q.comm, parts, id, q.id);
                if (s != 0) st = s;
            }
            if (group) { const int s = r.GroupEnd(); if (s != 0) st = s; }
            if (st != 0) { fprintf(stderr, "mspmv_mg: ncclCommInitRank failed (%d)\n", st); return fail(kErrRccl); }
        }
    }
    for (Part &q : plan->local) { (void) hipSetDevice(q.device); if (hipStreamSynchronize(q.stream) != hipSuccess) return fail(kErrInvalid); }
    (void) hipSetDevice(prev_dev);
    *out = plan;
    return 0;
}

int mspmv_mg_plan_set_part(mspmv_mg_plan_t *plan, int32_t i, const void *d_values, const int32_t *d_local_row_offsets,
                           const int32_t *d_column_indices)
{
    if (!plan || i < 0 || i >= (int) plan->local.size() || !d_local_row_offsets) return kErrInvalid;
    Part &q = plan->local[(size_t) i];
    if (q.local_nnz > 0 && (!d_values || !d_column_indices)) return kErrInvalid;
    int prev = 0; MG_HIP(hipGetDevice(&prev));
    MG_HIP(hipSetDevice(q.device));
    q.values = d_values; q.offsets = d_local_row_offsets; q.cols = d_column_indices;
    // the part's tile coordinates depend on its row offsets only: found once, here
    CallExtra ex; ex.phase = PHASE_COORDS_ONLY;
    size_t tb = q.temp_bytes;
    int st;
    if (plan->value_bytes == 4)
        st = csrmv_call<float>(q.temp, &tb, nullptr, q.offsets, nullptr, nullptr, nullptr, q.local_rows, 0, q.local_nnz, 1.f, 0.f, false, q.stream, 0, ex);
    else
        st = csrmv_call<double>(q.temp, &tb, nullptr, q.offsets, nullptr, nullptr, nullptr, q.local_rows, 0, q.local_nnz, 1.0, 0.0, false, q.stream, 0, ex);
    if (st == 0) st = (int) hipStreamSynchronize(q.stream);
    q.attached = st == 0;
    (void) hipSetDevice(prev);
    return st;
}

int mspmv_mg_plan_hot_columns(mspmv_mg_plan_t *plan, int32_t enable)
{
    if (!plan) return kErrInvalid;
    int prev = 0; MG_HIP(hipGetDevice(&prev));
    int st = 0;
    for (Part &q : plan->local) {
        if (!q.attached) { st = kErrInvalid; break; }
        (void) hipSetDevice(q.device);
        (void) hipStreamSynchronize(q.stream);
        if (!enable) { (void) hipFree(q.hot); q.hot = nullptr; q.hot_bytes = 0; continue; }
        if (q.hot) continue;
        size_t bytes = 0;
        st = mspmv_csrmv_hotcols_size(q.local_rows, (int32_t) plan->cols, q.local_nnz, plan->value_bytes, &bytes);
        if (st != 0) break;
        if (hipMalloc(&q.hot, bytes) != hipSuccess) { (void) hipGetLastError(); q.hot = nullptr; st = hipErrorOutOfMemory; break; }
        q.hot_bytes = bytes;
        st = mspmv_csrmv_hotcols_build(q.hot, bytes, q.offsets, q.cols, q.local_rows, (int32_t) plan->cols, q.local_nnz, plan->value_bytes, q.stream, 0);
        if (st == 0) st = (int) hipStreamSynchronize(q.stream);
        if (st != 0) { (void) hipFree(q.hot); q.hot = nullptr; q.hot_bytes = 0; break; }
    }
    (void) hipSetDevice(prev);
    return st;
}

void *mspmv_mg_plan_x(mspmv_mg_plan_t *plan, int32_t i)
{
    if (!plan || i < 0 || i >= (int) plan->local.size()) return nullptr;
    return plan->replicas[(size_t) plan->local[(size_t) i].replica].x;
}

void *mspmv_mg_plan_y(mspmv_mg_plan_t *plan, int32_t i)
{
    if (!plan || i < 0 || i >= (int) plan->local.size()) return nullptr;
    return plan->local[(size_t) i].y;
}

mspmv_stream_t mspmv_mg_plan_stream(mspmv_mg_plan_t *plan, int32_t i)
{
    if (!plan || i < 0 || i >= (int) plan->local.size()) return nullptr;
    return plan->local[(size_t) i].stream;
}

int mspmv_mg_plan_info(mspmv_mg_plan_t *plan, mspmv_mg_info_t *info)
{
    if (!plan || !info) return kErrInvalid;
    memset(info, 0, sizeof(*info));
    info->parts = plan->parts; info->local_parts = (int32_t) plan->local.size(); info->exchange = plan->exchange;
    info->value_bytes = plan->value_bytes; info->replicas = (int32_t) plan->replicas.size();
    info->rows = plan->rows; info->cols = plan->cols;
    info->carry_bytes_per_step = (uint64_t) plan->parts * (uint64_t) plan->value_bytes;
    // y -> x: every part's owned rows reach every OTHER replica once
    info->allgather_bytes_per_step = (uint64_t) plan->rows * (uint64_t) plan->value_bytes *
                                     (uint64_t) std::max<int>(plan->whole ? (int) plan->replicas.size() - 1 : plan->parts - 1, 0);
    info->steps = plan->steps;
    return 0;
}

int mspmv_mg_csrmv(mspmv_mg_plan_t *plan)
{
    if (!plan) return kErrInvalid;
    int prev = 0; MG_HIP(hipGetDevice(&prev));
    const int st = plan->value_bytes == 4 ? run_spmv<float>(plan) : run_spmv<double>(plan);
    (void) hipSetDevice(prev);
    return st;
}

int mspmv_mg_allgather_rows(mspmv_mg_plan_t *plan)
{
    if (!plan) return kErrInvalid;
    int prev = 0; MG_HIP(hipGetDevice(&prev));
    const int st = plan->value_bytes == 4 ? run_allgather<float>(plan) : run_allgather<double>(plan);
    (void) hipSetDevice(prev);
    return st;
}

int mspmv_mg_synchronize(mspmv_mg_plan_t *plan)
{
    if (!plan) return kErrInvalid;
    int prev = 0; MG_HIP(hipGetDevice(&prev));
    int st = 0;
    for (Part &q : plan->local) {
        (void) hipSetDevice(q.device);
        const hipError_t e = hipStreamSynchronize(q.stream);
        if (e != hipSuccess && st == 0) st = (int) e;
    }
    (void) hipSetDevice(prev);
    return st;
}

int mspmv_mg_plan_destroy(mspmv_mg_plan_t *plan)
{
    int prev = 0; (void) hipGetDevice(&prev);
    const int st = destroy(plan);
    (void) hipSetDevice(prev);
    return st;
}

}  // extern "C"
