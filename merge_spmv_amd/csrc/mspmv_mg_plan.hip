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
//   IPC   one process per GPU, no collective library: every process exports hipIpc handles of its x replica and of a small
//         mailbox block (mspmv_mg_plan_ipc_export), the launcher ships the blobs, every process opens its peers'
//         (mspmv_mg_plan_ipc_import).  A step is then the part's SpMV + ONE tiny kernel that writes the part's carry, tagged
//         with the step number, straight into the mailbox of the part that owns the row (a peer write over xGMI) + ONE
//         tiny kernel on the owner that waits for the tags of its sources, adds them in part order and acknowledges.  Step
//         tags instead of events or collectives: nothing on the host, no rendezvous; a producer may run at most two steps
//         ahead of its consumer (two slots per source, credit = the consumer's acknowledgement).  Waits are bounded.
// mspmv_mg_allgather_rows turns the row-sharded y into the replicated x of the next
// SpMV: PEER = every part pushes its owned rows straight into every replica of x (direct writes over the
// fully connected xGMI, unpadded, no host loop); RCCL = G grouped ncclBroadcast calls (the all-gather-v
// idiom), again unpadded and written in place; IPC = the same pushes into the peers' opened replicas, fenced by two
// step-tagged flags per pair ("my SpMV of step s has read x" before anyone overwrites it, "my rows of step s are in your
// x" before the next SpMV reads it).
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
    // the replicas may belong to other devices (and, in the hipIpc form, the "rows are there" flag is written by the NEXT
    // kernel of this stream and polled by a kernel of the owner's device): every element is stored with system scope --
    // written through, not left dirty in this device's L2 -- so the rows are visible system-wide when the kernel has ended,
    // whatever scope the end-of-kernel release of back-to-back launches has.  (A __threadfence_system() per thread instead
    // means one L2 write-back per wave: 7 x the time of the push, 0.014 -> 0.098 ms for 32 MB.)
    for (long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride)
        __hip_atomic_store(out + i, y[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- IPC backend: the mailbox block every part shares with its peers (device memory, opened by them through hipIpc) ----
constexpr int IPC_SLOTS = 2;                         // a producer may run this many steps ahead of its consumer
struct IpcBlock {
    unsigned long long carry[MSPMV_MG_MAX_PARTS][IPC_SLOTS][2];   // [source part][step & 1]: the source's carry as a record tagged with the step
    unsigned long long carry_ack[MSPMV_MG_MAX_PARTS];             // [consumer part]: the last step whose carry OF THIS PART that consumer has taken
    unsigned long long spmv_done[MSPMV_MG_MAX_PARTS];             // [part]: the last step whose SpMV that part has finished (it no longer reads its x)
    unsigned long long rows_pushed[MSPMV_MG_MAX_PARTS];           // [part]: the last step whose rows that part has written into THIS part's x
};
constexpr long long IPC_MAX_SPINS = 1LL << 24;       // x ~0.3 us: seconds; running out raises the part's error word (mspmv_mg_synchronize reports it)
__device__ __forceinline__ unsigned long long sys_load(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void sys_store(unsigned long long *p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ bool spin_until_at_least(const unsigned long long *p, unsigned long long want)
{
    for (long long i = 0; i < IPC_MAX_SPINS; ++i) { if (sys_load(p) >= want) return true; __builtin_amdgcn_s_sleep(8); }
    return false;
}
template <typename V> __device__ __forceinline__ unsigned long long value_bits(V v);
template <> __device__ __forceinline__ unsigned long long value_bits<float>(float v) { return __builtin_bit_cast(unsigned, v); }
template <> __device__ __forceinline__ unsigned long long value_bits<double>(double v) { return __builtin_bit_cast(unsigned long long, v); }
template <typename V> __device__ __forceinline__ V bits_value(unsigned long long b);
template <> __device__ __forceinline__ float bits_value<float>(unsigned long long b) { return __builtin_bit_cast(float, (unsigned) b); }
template <> __device__ __forceinline__ double bits_value<double>(unsigned long long b) { return __builtin_bit_cast(double, b); }

// What follows a part's SpMV in an IPC step, as ONE launch (three dependent tiny launches cost ~15 us of kernel boundaries
// per step): lanes 0..n_peers-1 announce "my SpMV of this step has read x" (done_flag[k]: spmv_done[me] in peer k's block);
// lane 0 then writes this part's carry of the step, as a record tagged with the step, into the mailboxes of the parts that
// own the row it belongs to (slot[k]: carry[me][step & 1] inside consumer k's block, peer memory; push_ack[k]: carry_ack[consumer k]
// inside MY block -- the credit: the consumer has taken what the slot held two steps ago) and, last, adds the carries of its
// sources to y_first[0] in part order and acknowledges them (rec[k]: carry[source k] inside MY block; take_ack[k]: carry_ack[me]
// inside source k's block).  Every part pushes before it waits: no cycle.
template <typename V>
__global__ __launch_bounds__(64) void ipc_step_tail_kernel(unsigned long long *const *__restrict__ done_flag, int n_peers,
                                                           const V *__restrict__ y_last, unsigned long long *const *__restrict__ slot,
                                                           const unsigned long long *const *__restrict__ push_ack, int n_push,
                                                           V *__restrict__ y_first, const unsigned long long *const *__restrict__ rec,
                                                           unsigned long long *const *__restrict__ take_ack, int n_take,
                                                           unsigned long long step, int *error)
{
    const int lane = threadIdx.x;
    for (int k = lane; k < n_peers; k += 64) sys_store(done_flag[k], step);
    if (lane != 0) return;
    const unsigned tag = (unsigned) step;
    if (n_push > 0) {
        const unsigned long long bits = value_bits<V>(*y_last);
        for (int k = 0; k < n_push; ++k) {
            if (step > IPC_SLOTS && !spin_until_at_least(push_ack[k], step - IPC_SLOTS)) { *error = 1; return; }
            unsigned long long *r = slot[k] + 2 * (step & (IPC_SLOTS - 1));
            sys_store(r, ((unsigned long long) tag << 32) | (unsigned) (bits >> 32));
            sys_store(r + 1, ((unsigned long long) tag << 32) | (unsigned) bits);
        }
    }
    if (n_take > 0) {
        V acc = *y_first;
        for (int k = 0; k < n_take; ++k) {
            const unsigned long long *r = rec[k] + 2 * (step & (IPC_SLOTS - 1));
            unsigned long long w0 = 0, w1 = 0; bool ok = false;
            for (long long i = 0; i < IPC_MAX_SPINS; ++i) {
                w0 = sys_load(r); w1 = sys_load(r + 1);
                if ((unsigned) (w0 >> 32) == tag && (unsigned) (w1 >> 32) == tag) { ok = true; break; }
                __builtin_amdgcn_s_sleep(8);
            }
            if (!ok) { *error = 2; acc = (V) __builtin_nan(""); break; }
            acc += bits_value<V>(((w0 & 0xffffffffull) << 32) | (w1 & 0xffffffffull));
            sys_store(take_ack[k], step);
        }
        *y_first = acc;
    }
}
// flag[k][me] = step in every peer's block (k = the other parts): "done" / "pushed" announcements
__global__ void ipc_announce_kernel(unsigned long long *const *__restrict__ flag, int n, unsigned long long step)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) sys_store(flag[k], step);
}
// wait until every listed counter of MY block has reached `step`
__global__ void ipc_wait_kernel(const unsigned long long *const *__restrict__ flag, int n, unsigned long long step, int *error)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n && !spin_until_at_least(flag[k], step)) *error = 3;
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
    hipEvent_t ex_begin = nullptr, ex_end = nullptr;    // timing events around the step's exchange alone (mspmv_mg_plan_exchange_ms)
    bool ex_valid = false;
    int hot_median = -1, hot_wide = 0;                  // what the automatic hot-column decision saw (mspmv_csrmv_hotcols_skew)
    ncclComm_t comm = nullptr;
    void *hot = nullptr; size_t hot_bytes = 0;      // hot-column plan of the part (mspmv_mg_plan_hot_columns): its own temp, indices, x
    // IPC backend
    IpcBlock *block = nullptr;            // this part's mailbox block (device memory, exported)
    int *ipc_error = nullptr;             // device word: a bounded wait ran out
    void **ipc_tables = nullptr;          // device array holding the pointer tables below, filled by mspmv_mg_plan_ipc_import
    int n_push = 0, n_take = 0, n_peers = 0, n_push_rows = 0;
    std::vector<int> consumers;           // parts that take this part's carry
};

struct Replica { int device = 0; void *x = nullptr; };

}  // namespace

struct IpcPeer { int part = -1; IpcBlock *block = nullptr; void *x = nullptr; bool opened = false; };

struct mspmv_mg_plan {
    std::vector<IpcPeer> peers;           // IPC backend: every part of the job (own parts: local pointers)
    bool ipc_ready = false;
    unsigned long long allgather_step = 0;    // IPC: the last step whose rows were pushed (the next SpMV waits for the peers' pushes)
    int parts = 0, value_bytes = 0, exchange = 0;
    long long rows = 0, cols = 0;
    bool whole = false;                   // this process holds every part
    std::vector<long long> row_split, nz_split;
    std::vector<Part> local;
    std::vector<Replica> replicas;
    unsigned long long steps = 0;
    int hot_mode = -1;                    // mspmv_mg_plan_hot_columns: -1 automatic (default), 0 never, 1 always
};

static int part_hot(mspmv_mg_plan *plan, Part &q, bool want);
static int part_hot_auto(mspmv_mg_plan *plan, Part &q);

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
    const bool ipc = plan->exchange == MSPMV_MG_EXCHANGE_IPC;
    if (ipc && !plan->ipc_ready) return kErrInvalid;
    const unsigned long long step = plan->steps + 1;
    for (Part &q : plan->local) {
        if (!q.attached) return kErrInvalid;
        MG_HIP(hipSetDevice(q.device));
        if (ipc && q.n_peers > 0 && plan->allgather_step + 1 == step && step > 1) {
            // the peers' rows of the previous step must be in this part's x before the SpMV reads it
            hipLaunchKernelGGL(ipc_wait_kernel, dim3(1), dim3(64), 0, q.stream,
                               reinterpret_cast<const unsigned long long *const *>(q.ipc_tables + 7 * MSPMV_MG_MAX_PARTS), q.n_peers, step - 1, q.ipc_error);
        }
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
        MG_HIP(hipEventRecord(q.ex_begin, q.stream));       // (the exchange of this step begins here on this part's stream ...)
        if (ipc) {
            const int vb = (int) sizeof(V);
            // "my SpMV of this step has read x" -> every peer (a later row all-gather waits for it before overwriting x); this
            // part's carry -> its consumers' mailboxes; the carries of its sources -> y[0]: one launch
            if (q.n_peers > 0 || q.n_push > 0 || q.n_take > 0)
                hipLaunchKernelGGL((ipc_step_tail_kernel<V>), dim3(1), dim3(64), 0, q.stream,
                                   reinterpret_cast<unsigned long long *const *>(q.ipc_tables + 4 * MSPMV_MG_MAX_PARTS), q.n_peers,
                                   reinterpret_cast<const V *>(static_cast<const char *>(q.y) + (size_t) (q.local_rows - 1) * vb),
                                   reinterpret_cast<unsigned long long *const *>(q.ipc_tables + 0 * MSPMV_MG_MAX_PARTS),
                                   reinterpret_cast<const unsigned long long *const *>(q.ipc_tables + 1 * MSPMV_MG_MAX_PARTS), q.n_push,
                                   static_cast<V *>(q.y),
                                   reinterpret_cast<const unsigned long long *const *>(q.ipc_tables + 2 * MSPMV_MG_MAX_PARTS),
                                   reinterpret_cast<unsigned long long *const *>(q.ipc_tables + 3 * MSPMV_MG_MAX_PARTS), q.n_take, step, q.ipc_error);
            MG_HIP(hipGetLastError());
            MG_HIP(hipEventRecord(q.ex_end, q.stream)); q.ex_valid = true;
        }
    }
    if (ipc) { ++plan->steps; return 0; }
    if (plan->parts == 1 && plan->exchange != MSPMV_MG_EXCHANGE_RCCL) {
        for (Part &q : plan->local) { MG_HIP(hipSetDevice(q.device)); MG_HIP(hipEventRecord(q.ex_end, q.stream)); q.ex_valid = true; }
        ++plan->steps; return 0;
    }
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
    // (... and ends here: the all-gather / the peers' events and the owner's add, nothing else)
    for (Part &q : plan->local) { MG_HIP(hipSetDevice(q.device)); MG_HIP(hipEventRecord(q.ex_end, q.stream)); q.ex_valid = true; }
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
    if (plan->exchange == MSPMV_MG_EXCHANGE_IPC) {
        if (!plan->ipc_ready || plan->steps == 0) return kErrInvalid;
        const unsigned long long step = plan->steps;              // the step whose y becomes x
        for (Part &q : plan->local) {
            MG_HIP(hipSetDevice(q.device));
            // nobody may still be reading x: every peer has announced its SpMV of this step
            if (q.n_peers > 0)
                hipLaunchKernelGGL(ipc_wait_kernel, dim3(1), dim3(64), 0, q.stream,
                                   reinterpret_cast<const unsigned long long *const *>(q.ipc_tables + 6 * MSPMV_MG_MAX_PARTS), q.n_peers, step, q.ipc_error);
            if (q.owned > 0) {
                const unsigned grid = (unsigned) std::min<long long>((q.owned + 255) / 256, 1024);
                hipLaunchKernelGGL((mg_push_rows_kernel<V>), dim3(grid, (unsigned) q.n_push_rows), dim3(256), 0, q.stream,
                                   static_cast<const V *>(q.y), reinterpret_cast<V *const *>(q.push_table), q.owned);
            }
            // (a kernel boundary on this stream: the rows are written before the announcement goes out)
            if (q.n_peers > 0)
                hipLaunchKernelGGL(ipc_announce_kernel, dim3(1), dim3(64), 0, q.stream,
                                   reinterpret_cast<unsigned long long *const *>(q.ipc_tables + 5 * MSPMV_MG_MAX_PARTS), q.n_peers, step);
            MG_HIP(hipGetLastError());
        }
        plan->allgather_step = step;
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
        if (q.ex_begin) (void) hipEventDestroy(q.ex_begin);
        if (q.ex_end) (void) hipEventDestroy(q.ex_end);
        if (q.applied) (void) hipEventDestroy(q.applied);
        if (q.pushed) (void) hipEventDestroy(q.pushed);
        (void) hipFree(q.temp); (void) hipFree(q.y); (void) hipFree(q.carries); (void) hipFree(q.src_table); (void) hipFree(q.push_table);
        (void) hipFree(q.hot); (void) hipFree(q.block); (void) hipFree(q.ipc_error); (void) hipFree(q.ipc_tables);
        if (q.stream) (void) hipStreamDestroy(q.stream);
    }
    for (IpcPeer &peer : plan->peers)
        if (peer.opened) { if (peer.block) (void) hipIpcCloseMemHandle(peer.block); if (peer.x) (void) hipIpcCloseMemHandle(peer.x); }
    (void) hipGetLastError();
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
        exchange > MSPMV_MG_EXCHANGE_IPC)
        return kErrInvalid;
    *out = nullptr;
    int ndev = 0, prev_dev = 0;
    MG_HIP(hipGetDeviceCount(&ndev));
    MG_HIP(hipGetDevice(&prev_dev));
    mspmv_mg_plan *plan = new mspmv_mg_plan;
    plan->parts = parts; plan->value_bytes = value_bytes; plan->cols = cols; plan->rows = row_split[parts];
    plan->whole = local_parts == parts && !id128 && exchange != MSPMV_MG_EXCHANGE_IPC;     // (an id / the IPC backend make even a 1-rank job take the multi-process path)
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
    if (exchange == MSPMV_MG_EXCHANGE_PEER && local_parts != parts) return fail(kErrInvalid);       // peers live in this process (other processes: MSPMV_MG_EXCHANGE_IPC)
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
            hipEventCreateWithFlags(&q.pushed, hipEventDisableTiming) != hipSuccess ||
            hipEventCreate(&q.ex_begin) != hipSuccess || hipEventCreate(&q.ex_end) != hipSuccess)
            return fail(hipErrorOutOfMemory);
        if (exchange == MSPMV_MG_EXCHANGE_RCCL) {
            if (hipMalloc(&q.carries, std::max<size_t>((size_t) parts * vb, 256)) != hipSuccess) return fail(hipErrorOutOfMemory);
            if (hipMemsetAsync(q.carries, 0, (size_t) parts * vb, q.stream) != hipSuccess) return fail(kErrInvalid);
        }
    }
    if (exchange == MSPMV_MG_EXCHANGE_IPC)
        for (Part &q : plan->local) {
            if (hipSetDevice(q.device) != hipSuccess) return fail(kErrInvalid);
            // the mailbox is polled by kernels of THIS device while kernels of other devices write it over the links: ordinary
            // (coarse-grained) device memory is only coherent between agents at kernel boundaries, so the block is allocated
            // uncached (else fine-grained).  No plain allocation as a last resort: it would only be right with every part on one
            // device, which a plan that sees its own parts alone cannot know
            void *blk = nullptr;
            if (hipExtMallocWithFlags(&blk, sizeof(IpcBlock), hipDeviceMallocUncached) != hipSuccess) {
                (void) hipGetLastError(); blk = nullptr;
                if (hipExtMallocWithFlags(&blk, sizeof(IpcBlock), hipDeviceMallocFinegrained) != hipSuccess) {
                    (void) hipGetLastError();
                    return fail(hipErrorNotSupported);
                }
            }
            q.block = static_cast<IpcBlock *>(blk);
            if (hipMalloc(reinterpret_cast<void **>(&q.ipc_error), 256) != hipSuccess ||
                hipMalloc(reinterpret_cast<void **>(&q.ipc_tables), sizeof(void *) * 8 * MSPMV_MG_MAX_PARTS) != hipSuccess)
                return fail(hipErrorOutOfMemory);
            if (hipMemset(q.block, 0, sizeof(IpcBlock)) != hipSuccess || hipMemset(q.ipc_error, 0, 256) != hipSuccess) return fail(kErrInvalid);
            // who takes this part's carry: the first later part that owns a row, if that row is the one this part leaves open
            for (int c = q.id + 1; c < parts; ++c)
                if (row_split[c + 1] > row_split[c]) { if (row_split[c] == row_split[q.id + 1]) q.consumers.push_back(c); break; }
        }
    // address tables (all buffers they point into are plan-owned, so they are fixed for the plan's life)
    for (Part &q : plan->local) {
        if (hipSetDevice(q.device) != hipSuccess) return fail(kErrInvalid);
        if (q.nsrc > 0 && exchange != MSPMV_MG_EXCHANGE_IPC) {
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
    // a new matrix: whatever hot-column plan the part had belonged to the old one; the plan's mode decides about a new one
    if (q.attached) { (void) part_hot(plan, q, false); st = plan->hot_mode < 0 ? part_hot_auto(plan, q) : plan->hot_mode ? part_hot(plan, q, true) : 0; }
    (void) hipSetDevice(prev);
    return st;
}

// ---- IPC backend: handle exchange.  A blob = int32 count, then per local part { int32 part, int32 device, 64-byte handle of the
// mailbox block, 64-byte handle of the x replica }.
struct IpcEntry { int32_t part, device; hipIpcMemHandle_t block, x; };
static_assert(sizeof(hipIpcMemHandle_t) == 64, "blob layout");

int mspmv_mg_plan_ipc_export(mspmv_mg_plan_t *plan, void *blob, size_t *blob_bytes)
{
    if (!plan || !blob_bytes || plan->exchange != MSPMV_MG_EXCHANGE_IPC) return kErrInvalid;
    const size_t need = 8 + plan->local.size() * sizeof(IpcEntry);
    if (!blob) { *blob_bytes = need; return 0; }
    if (*blob_bytes < need) return kErrInvalid;
    int prev = 0; MG_HIP(hipGetDevice(&prev));
    char *out = static_cast<char *>(blob);
    const int32_t n = (int32_t) plan->local.size();
    memset(out, 0, need); memcpy(out, &n, 4);
    for (size_t i = 0; i < plan->local.size(); ++i) {
        Part &q = plan->local[i];
        MG_HIP(hipSetDevice(q.device));
        IpcEntry e; memset(&e, 0, sizeof(e)); e.part = q.id; e.device = q.device;
        MG_HIP(hipIpcGetMemHandle(&e.block, q.block));
        MG_HIP(hipIpcGetMemHandle(&e.x, plan->replicas[q.replica].x));
        memcpy(out + 8 + i * sizeof(IpcEntry), &e, sizeof(e));
    }
    (void) hipSetDevice(prev);
    *blob_bytes = need;
    return 0;
}

int mspmv_mg_plan_ipc_import(mspmv_mg_plan_t *plan, const void *blobs, int32_t count, size_t blob_stride)
{
    if (!plan || !blobs || count < 1 || plan->exchange != MSPMV_MG_EXCHANGE_IPC || plan->ipc_ready) return kErrInvalid;
    int prev = 0; MG_HIP(hipGetDevice(&prev));
    plan->peers.assign((size_t) plan->parts, IpcPeer());
    for (Part &q : plan->local) { IpcPeer &me = plan->peers[(size_t) q.id]; me.part = q.id; me.block = q.block; me.x = plan->replicas[q.replica].x; }
    struct Opened { hipIpcMemHandle_t h; void *ptr; };
    std::vector<Opened> opened;                      // (two parts of one remote process may share an x replica: open each handle once)
    auto open_handle = [&](const hipIpcMemHandle_t &h, void **out) -> int {
        for (const Opened &o : opened) if (memcmp(&o.h, &h, sizeof(h)) == 0) { *out = o.ptr; return 0; }
        void *ptr = nullptr;
        const hipError_t e = hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) { (void) hipGetLastError(); return (int) e; }
        opened.push_back(Opened{h, ptr}); *out = ptr;
        return 0;
    };
    // the handles are opened with ONE device current (peer access is enabled for that device): a process whose local parts sit on
    // several devices would leave the others without access to the imported memory -- refused rather than half-working
    for (const Part &q : plan->local) if (q.device != plan->local[0].device) return kErrInvalid;
    MG_HIP(hipSetDevice(plan->local[0].device));
    // on any failure: whatever was opened so far is closed again and the plan is left as it was (a retry starts clean)
    auto give_up = [&](int st) -> int {
        for (const Opened &o : opened) (void) hipIpcCloseMemHandle(o.ptr);
        plan->peers.clear();
        (void) hipSetDevice(prev);
        return st;
    };
    for (int b = 0; b < count; ++b) {
        const char *in = static_cast<const char *>(blobs) + (size_t) b * blob_stride;
        int32_t n = 0; memcpy(&n, in, 4);
        if (n < 0 || 8 + (size_t) n * sizeof(IpcEntry) > blob_stride) return give_up(kErrInvalid);
        for (int i = 0; i < n; ++i) {
            IpcEntry e; memcpy(&e, in + 8 + (size_t) i * sizeof(IpcEntry), sizeof(e));
            if (e.part < 0 || e.part >= plan->parts) return give_up(kErrInvalid);
            IpcPeer &peer = plan->peers[(size_t) e.part];
            if (peer.part == e.part) continue;                       // one of this process's own parts
            void *blk = nullptr, *x = nullptr;
            int st = open_handle(e.block, &blk);
            if (st == 0) st = open_handle(e.x, &x);
            if (st != 0) return give_up(st);
            peer.part = e.part; peer.block = static_cast<IpcBlock *>(blk); peer.x = x; peer.opened = true;
        }
    }
    for (const IpcPeer &peer : plan->peers) if (peer.part < 0) return give_up(kErrInvalid);     // a part nobody exported
    // every peer's pointers are known: the tables of the step kernels
    const size_t vb = (size_t) plan->value_bytes;
    for (Part &q : plan->local) {
        MG_HIP(hipSetDevice(q.device));
        std::vector<void *> t((size_t) 8 * MSPMV_MG_MAX_PARTS, nullptr);
        auto at = [&](int table, int k) -> void *& { return t[(size_t) table * MSPMV_MG_MAX_PARTS + (size_t) k]; };
        q.n_push = 0;
        for (int c : q.consumers) {
            at(0, q.n_push) = &plan->peers[(size_t) c].block->carry[q.id][0][0];         // where my carry goes
            at(1, q.n_push) = &q.block->carry_ack[c];                                    // where the consumer acknowledges
            ++q.n_push;
        }
        q.n_take = 0;
        for (int src : q.sources) {
            at(2, q.n_take) = &q.block->carry[src][0][0];
            at(3, q.n_take) = &plan->peers[(size_t) src].block->carry_ack[q.id];
            ++q.n_take;
        }
        q.n_peers = 0;
        for (int o = 0; o < plan->parts; ++o) {
            if (o == q.id) continue;
            at(4, q.n_peers) = &plan->peers[(size_t) o].block->spmv_done[q.id];
            at(5, q.n_peers) = &plan->peers[(size_t) o].block->rows_pushed[q.id];
            at(6, q.n_peers) = &q.block->spmv_done[o];
            at(7, q.n_peers) = &q.block->rows_pushed[o];
            ++q.n_peers;
        }
        MG_HIP(hipMemcpy(q.ipc_tables, t.data(), t.size() * sizeof(void *), hipMemcpyHostToDevice));
        // row pushes: this part's row range inside every distinct replica of x in the job
        std::vector<void *> dst;
        for (const IpcPeer &peer : plan->peers) {
            void *d = static_cast<char *>(peer.x) + (size_t) q.row_begin * vb;
            if (std::find(dst.begin(), dst.end(), d) == dst.end()) dst.push_back(d);
        }
        q.n_push_rows = (int) dst.size();
        if (q.push_table) { (void) hipFree(q.push_table); q.push_table = nullptr; }
        MG_HIP(hipMalloc(reinterpret_cast<void **>(&q.push_table), dst.size() * sizeof(void *)));
        MG_HIP(hipMemcpy(q.push_table, dst.data(), dst.size() * sizeof(void *), hipMemcpyHostToDevice));
    }
    (void) hipSetDevice(prev);
    plan->ipc_ready = true;
    return 0;
}

// the Infinity Cache the automatic hot-column decision compares x with (dev library only: MSPMV_FAKE_INFINITY_CACHE_MIB in the
// environment overrides it, read once: tests)
static long long infinity_cache_bytes()
{
#ifdef MSPMV_TUNING
    static const long long v = [] { const char *e = getenv("MSPMV_FAKE_INFINITY_CACHE_MIB"); const double m = e ? atof(e) : 0; return m > 0 ? (long long) (m * 1048576.0) : (256LL << 20); }();
    return v;
#else
    return 256LL << 20;
#endif
}

// one part's hot-column plan: built (want) or dropped; the part's stream is synchronised on the way
static int part_hot(mspmv_mg_plan *plan, Part &q, bool want)
{
    (void) hipSetDevice(q.device);
    (void) hipStreamSynchronize(q.stream);
    if (!want) { (void) hipFree(q.hot); q.hot = nullptr; q.hot_bytes = 0; return 0; }
    if (q.hot) return 0;
    size_t bytes = 0;
    int st = mspmv_csrmv_hotcols_size(q.local_rows, (int32_t) plan->cols, q.local_nnz, plan->value_bytes, &bytes);
    if (st != 0) return st;
    if (hipMalloc(&q.hot, bytes) != hipSuccess) { (void) hipGetLastError(); q.hot = nullptr; return hipErrorOutOfMemory; }
    q.hot_bytes = bytes;
    st = mspmv_csrmv_hotcols_build(q.hot, bytes, q.offsets, q.cols, q.local_rows, (int32_t) plan->cols, q.local_nnz, plan->value_bytes, q.stream, 0);
    if (st == 0) st = (int) hipStreamSynchronize(q.stream);
    if (st != 0) { (void) hipFree(q.hot); q.hot = nullptr; q.hot_bytes = 0; }
    return st;
}

// AUTOMATIC (the default): a part whose x replica is beyond the Infinity Cache -- every gather that misses the L2s then moves a
// 128-byte line from DRAM, 55 G lines/s on MI355X whatever the kernel does -- and whose columns come back (a scale-free matrix:
// a sample of ~1 M references touches 15 .. 80 % of the distinct lines a uniform draw would, and spans x; mspmv_csrmv_hotcols_skew) gets the
// plan; a part of uniformly spread columns (nothing to concentrate), of a stencil or a band (already cache-friendly), or with an x
// that fits the cache does not.  A part that cannot afford the plan's memory runs without it.  y is bit for bit the same either way.
static int part_hot_auto(mspmv_mg_plan *plan, Part &q)
{
    q.hot_median = -1; q.hot_wide = 0;
    if ((long long) plan->cols * plan->value_bytes <= infinity_cache_bytes() || q.local_nnz < 2048) return part_hot(plan, q, false);
    (void) hipSetDevice(q.device);
    int32_t median = -1, wide = 0;
    const int st = mspmv_csrmv_hotcols_skew(q.cols, (int32_t) plan->cols, q.local_nnz, plan->value_bytes, q.stream, &median, &wide);
    if (st != 0) return st;
    q.hot_median = median; q.hot_wide = wide;
    const bool want = median >= 150 && median < 800 && wide >= 256;        // (distinct lines per mille of a uniform draw; 512 windows: profiles/r05_skew_probe.txt)
    const int built = part_hot(plan, q, want);
    return built == (int) hipErrorOutOfMemory ? 0 : built;
}

int mspmv_mg_plan_hot_columns(mspmv_mg_plan_t *plan, int32_t enable)
{
    if (!plan) return kErrInvalid;
    int prev = 0; MG_HIP(hipGetDevice(&prev));
    int st = 0;
    // the mode holds for every matrix attached from now on (mspmv_mg_plan_set_part applies it); parts that already have their
    // matrix are brought to it here, parts still waiting for one are left alone
    plan->hot_mode = enable < 0 ? -1 : enable ? 1 : 0;
    for (Part &q : plan->local) {
        if (!q.attached) continue;
        st = enable < 0 ? part_hot_auto(plan, q) : part_hot(plan, q, enable != 0);
        if (st != 0) break;
    }
    (void) hipSetDevice(prev);
    return st;
}

int mspmv_mg_plan_exchange_ms(mspmv_mg_plan_t *plan, int32_t i, float *ms)
{
    if (!plan || !ms || i < 0 || i >= (int) plan->local.size()) return kErrInvalid;
    Part &q = plan->local[(size_t) i];
    if (!q.ex_valid) return kErrInvalid;
    int prev = 0; MG_HIP(hipGetDevice(&prev));
    (void) hipSetDevice(q.device);
    hipError_t e = hipEventSynchronize(q.ex_end);
    if (e == hipSuccess) e = hipEventElapsedTime(ms, q.ex_begin, q.ex_end);
    (void) hipSetDevice(prev);
    return (int) e;
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
    for (const Part &q : plan->local) info->hot_parts += q.hot ? 1 : 0;
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
        if (q.ipc_error && st == 0) {
            int h = 0;
            if (hipMemcpy(&h, q.ipc_error, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess && h != 0) {
                fprintf(stderr, "mspmv_mg: part %d gave up waiting for a peer (code %d: 1 carry credit, 2 carry, 3 row all-gather flag)\n", q.id, h);
                st = hipErrorLaunchFailure;
            }
        }
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
