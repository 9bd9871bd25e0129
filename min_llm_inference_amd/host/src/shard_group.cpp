// include/mli_shard.h: N engines of one process, one per GPU and host thread, stepped in lock step with one RCCL
// all-gather of the generated token ids per rank and iteration.  The reference has nothing to mirror here
// (README.md:84-86 is a plan); the engines are its start_paged_attention_*_inference_engine loops (include/inferencer.h:18-32)
// behind include/mli_engine.h.  RCCL is opened with dlopen on first use: single-GPU users never load it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "mli_shard.h"

namespace {

thread_local std::string g_shard_error;

struct Rccl {
    decltype(&ncclCommInitAll) comm_init_all = nullptr;
    decltype(&ncclCommDestroy) comm_destroy = nullptr;
    decltype(&ncclAllGather) all_gather = nullptr;
    decltype(&ncclAllReduce) all_reduce = nullptr;
    decltype(&ncclGroupStart) group_start = nullptr;
    decltype(&ncclGroupEnd) group_end = nullptr;
    decltype(&ncclGetErrorString) error_string = nullptr;
};

// resolved once per process; throws when the library or a symbol is missing (there is no fallback collective)
const Rccl& rccl() {
    static Rccl api;
    static std::once_flag once;
    static std::string failure;
    std::call_once(once, [] {
        void* h = nullptr;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (h) break;
        }
        if (!h) {
            failure = std::string("librccl.so cannot be opened: ") + dlerror();
            return;
        }
        auto sym = [&](const char* n) {
            void* p = dlsym(h, n);
            if (!p && failure.empty()) failure = std::string("librccl.so lacks ") + n;
            return p;
        };
        api.comm_init_all = reinterpret_cast<decltype(api.comm_init_all)>(sym("ncclCommInitAll"));
        api.comm_destroy = reinterpret_cast<decltype(api.comm_destroy)>(sym("ncclCommDestroy"));
        api.all_gather = reinterpret_cast<decltype(api.all_gather)>(sym("ncclAllGather"));
        api.all_reduce = reinterpret_cast<decltype(api.all_reduce)>(sym("ncclAllReduce"));
        api.group_start = reinterpret_cast<decltype(api.group_start)>(sym("ncclGroupStart"));
        api.group_end = reinterpret_cast<decltype(api.group_end)>(sym("ncclGroupEnd"));
        api.error_string = reinterpret_cast<decltype(api.error_string)>(sym("ncclGetErrorString"));
    });
    if (!failure.empty()) throw std::runtime_error(failure);
    return api;
}

void hip_ok(hipError_t e, const char* what) {
    if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}
void nccl_ok(ncclResult_t r, const char* what) {
    if (r != ncclSuccess) throw std::runtime_error(std::string(what) + ": " + rccl().error_string(r));
}

// a reusable barrier for the rank threads that also ORs a flag ("somebody failed", "somebody still has work")
class StepBarrier {
public:
    explicit StepBarrier(int n) : n_(n) {}
    // returns the OR of every participant's `flag` of this round
    bool arrive(bool flag) {
        std::unique_lock<std::mutex> lock(m_);
        acc_ = acc_ || flag;
        const unsigned long long gen = gen_;
        if (++count_ == n_) {
            result_ = acc_;
            acc_ = false;
            count_ = 0;
            ++gen_;
            cv_.notify_all();
            return result_;
        }
        cv_.wait(lock, [&] { return gen_ != gen; });
        return result_;
    }

private:
    std::mutex m_;
    std::condition_variable cv_;
    int n_, count_ = 0;
    unsigned long long gen_ = 0;
    bool acc_ = false, result_ = false;
};

}  // namespace

struct mli_shard_group {
    int n = 0;
    std::vector<int> devices;
    std::vector<mli_engine*> engines;
    std::vector<ncclComm_t> comms;
    std::vector<void*> gathered;   // per rank: int32 [n][count] on its device
    std::vector<void*> streams;
    int count = 0;                 // token ids per rank and iteration
    int ranks_seen = 0;
    bool ran = false;
    bool loopback = false;         // every rank on one device; the exchange is device-to-device copies, no communicator

    ~mli_shard_group() {
        for (int r = 0; r < (int)comms.size(); ++r)
            if (comms[r] && !loopback) {
                (void)hipSetDevice(devices[r]);
                rccl().comm_destroy(comms[r]);
            }
        for (int r = 0; r < (int)gathered.size(); ++r)
            if (gathered[r]) {
                (void)hipSetDevice(devices[r]);
                (void)hipFree(gathered[r]);
            }
        for (mli_engine* e : engines) mli_engine_destroy(e);
    }
};

#define SHARD_GUARD(body)                          \
    try {                                          \
        body;                                      \
        return 0;                                  \
    } catch (const std::exception& e) {            \
        g_shard_error = e.what();                  \
        return -1;                                 \
    } catch (...) {                                \
        g_shard_error = "unknown C++ exception";   \
        return -1;                                 \
    }

extern "C" {

const char* mli_shard_last_error(void) { return g_shard_error.c_str(); }

}  // extern "C"

namespace {

// the engines, their private streams and the gathered buffers of a group (no exchange yet)
std::unique_ptr<mli_shard_group> make_group(const mli_engine_config* config, int n_ranks, const int* devices,
                                            const float* emb_table, const float* pos_table, const float* wk, const float* wq,
                                            const float* wv) {
        auto group = std::unique_ptr<mli_shard_group>(new mli_shard_group());
        group->n = n_ranks;
        group->devices.assign(devices, devices + n_ranks);
        group->comms.assign(n_ranks, nullptr);
        group->gathered.assign(n_ranks, nullptr);
        group->streams.assign(n_ranks, nullptr);
        for (int r = 0; r < n_ranks; ++r) {
            mli_engine_config c = *config;
            c.device = devices[r];
            mli_engine* e = nullptr;
            if (mli_engine_create(&c, emb_table, pos_table, wk, wq, wv, &e) != 0)
                throw std::runtime_error(std::string("rank engine: ") + mli_engine_last_error());
            group->engines.push_back(e);
            // lock-step stepping = the reference's sequential loop order; a private stream so that the ranks' threads overlap
            if (mli_engine_set_pipelined(e, 0) != 0 || mli_engine_use_private_stream(e) != 0 ||
                mli_engine_stream(e, &group->streams[r]) != 0)
                throw std::runtime_error(std::string("rank engine: ") + mli_engine_last_error());
            void* result = nullptr;
            int count = 0;
            if (mli_engine_decoder_result(e, &result, &count) != 0) throw std::runtime_error(mli_engine_last_error());
            group->count = count;
            hip_ok(hipSetDevice(devices[r]), "hipSetDevice");
            hip_ok(hipMalloc(&group->gathered[r], (size_t)n_ranks * count * sizeof(int)), "hipMalloc(gathered)");
            hip_ok(hipMemset(group->gathered[r], 0, (size_t)n_ranks * count * sizeof(int)), "hipMemset(gathered)");
        }
        return group;
}

}  // namespace

extern "C" {

int mli_shard_group_create_loopback(const mli_engine_config* config, int n_ranks, int device, const float* emb_table,
                                    const float* pos_table, const float* wk, const float* wq, const float* wv,
                                    mli_shard_group** out) {
    if (!config || !out || n_ranks < 1 || n_ranks > 64) { g_shard_error = "bad argument"; return -1; }
    SHARD_GUARD({
        const std::vector<int> devices(n_ranks, device);
        auto group = make_group(config, n_ranks, devices.data(), emb_table, pos_table, wk, wq, wv);
        group->loopback = true;
        group->ranks_seen = n_ranks;
        *out = group.release();
    })
}

int mli_shard_group_create(const mli_engine_config* config, int n_ranks, const int* devices, const float* emb_table,
                           const float* pos_table, const float* wk, const float* wq, const float* wv,
                           mli_shard_group** out) {
    if (!config || !devices || !out || n_ranks < 1 || n_ranks > 64) { g_shard_error = "bad argument"; return -1; }
    for (int a = 0; a < n_ranks; ++a)
        for (int b = a + 1; b < n_ranks; ++b)
            if (devices[a] == devices[b]) { g_shard_error = "one rank per device: duplicate device ordinal"; return -1; }
    SHARD_GUARD({
        const Rccl& api = rccl();
        auto group = make_group(config, n_ranks, devices, emb_table, pos_table, wk, wq, wv);
        nccl_ok(api.comm_init_all(group->comms.data(), n_ranks, group->devices.data()), "ncclCommInitAll");
        // self-proof that the communicator spans n_ranks: all-reduce of ones (grouped: one thread drives every rank here),
        // in the first word of every rank's gathered buffer (owned by the group: nothing to leak on the error paths)
        const int one = 1;
        const int zero = 0;
        for (int r = 0; r < n_ranks; ++r) {
            hip_ok(hipSetDevice(devices[r]), "hipSetDevice");
            hip_ok(hipMemcpy(group->gathered[r], &one, sizeof(int), hipMemcpyHostToDevice), "hipMemcpy");
        }
        nccl_ok(api.group_start(), "ncclGroupStart");
        for (int r = 0; r < n_ranks; ++r)
            nccl_ok(api.all_reduce(group->gathered[r], group->gathered[r], 1, ncclInt32, ncclSum, group->comms[r],
                                   static_cast<hipStream_t>(group->streams[r])), "ncclAllReduce");
        nccl_ok(api.group_end(), "ncclGroupEnd");
        for (int r = 0; r < n_ranks; ++r) {
            hip_ok(hipSetDevice(devices[r]), "hipSetDevice");
            hip_ok(hipStreamSynchronize(static_cast<hipStream_t>(group->streams[r])), "hipStreamSynchronize");
            int seen = 0;
            hip_ok(hipMemcpy(&seen, group->gathered[r], sizeof(int), hipMemcpyDeviceToHost), "hipMemcpy");
            hip_ok(hipMemcpy(group->gathered[r], &zero, sizeof(int), hipMemcpyHostToDevice), "hipMemcpy");
            if (seen != n_ranks) throw std::runtime_error("the communicator does not span every rank");
            group->ranks_seen = seen;
        }
        *out = group.release();
    })
}

void mli_shard_group_destroy(mli_shard_group* group) { delete group; }

int mli_shard_group_size(const mli_shard_group* group) { return group ? group->n : 0; }

int mli_shard_group_add_item(mli_shard_group* group, int id, const int* tokens, int n_tokens) {
    if (!group) { g_shard_error = "null group"; return -1; }
    const int rank = (int)((unsigned)id % (unsigned)group->n);
    if (mli_engine_add_item(group->engines[rank], id, tokens, n_tokens) != 0) {
        g_shard_error = mli_engine_last_error();
        return -1;
    }
    return 0;
}

int mli_shard_group_run(mli_shard_group* group, mli_shard_stats* stats) {
    if (!group) { g_shard_error = "null group"; return -1; }
    SHARD_GUARD({
        if (group->ran) throw std::runtime_error("a shard group runs once");
        group->ran = true;
        const Rccl* api = group->loopback ? nullptr : &rccl();
        const int n = group->n;
        StepBarrier barrier(n);
        std::vector<std::string> errors(n);
        std::vector<long long> iterations(n, 0);
        std::vector<double> gather_seconds(n, 0.0);
        std::atomic<bool> failed{false};
        const auto t_begin = std::chrono::steady_clock::now();
        auto rank_main = [&](int r) {
            bool done = false;
            try {
                hip_ok(hipSetDevice(group->devices[r]), "hipSetDevice");
            } catch (const std::exception& e) {
                errors[r] = e.what();
                failed.store(true);
            }
            void* result = nullptr;
            int count = 0;
            (void)mli_engine_decoder_result(group->engines[r], &result, &count);
            for (;;) {
                // every thread takes the same decision in every round: a failed rank keeps arriving at the barrier and the
                // round that sees the failure is the last for all (nobody is left waiting inside a collective)
                if (!failed.load() && !done) {
                    int d = 0;
                    if (mli_engine_step(group->engines[r], &d) != 0) {
                        errors[r] = mli_engine_last_error();
                        failed.store(true);
                    }
                    done = d != 0;
                }
                if (barrier.arrive(failed.load())) break;   // somebody failed before the collective of this round
                const auto g0 = std::chrono::steady_clock::now();
                if (api) {
                    const ncclResult_t rc = api->all_gather(result, group->gathered[r], (size_t)count, ncclInt32, group->comms[r],
                                                            static_cast<hipStream_t>(group->streams[r]));
                    if (rc != ncclSuccess) {
                        errors[r] = std::string("ncclAllGather: ") + api->error_string(rc);
                        failed.store(true);
                    }
                } else {   // loopback: this rank's slice into every rank's buffer, behind the forward on this rank's stream
                    for (int q = 0; q < n && errors[r].empty(); ++q) {
                        const hipError_t e = hipMemcpyAsync(static_cast<int*>(group->gathered[q]) + (size_t)r * count, result,
                                                            (size_t)count * sizeof(int), hipMemcpyDeviceToDevice,
                                                            static_cast<hipStream_t>(group->streams[r]));
                        if (e != hipSuccess) {
                            errors[r] = std::string("loopback gather: ") + hipGetErrorString(e);
                            failed.store(true);
                        }
                    }
                }
                gather_seconds[r] += std::chrono::duration<double>(std::chrono::steady_clock::now() - g0).count();
                ++iterations[r];
                if (!barrier.arrive(!done || failed.load())) break;   // nobody has work left
                if (failed.load()) break;
            }
            (void)hipStreamSynchronize(static_cast<hipStream_t>(group->streams[r]));
        };
        std::vector<std::thread> threads;
        for (int r = 1; r < n; ++r) threads.emplace_back(rank_main, r);
        rank_main(0);
        for (auto& t : threads) t.join();
        const double seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
        for (int r = 0; r < n; ++r)
            if (!errors[r].empty()) throw std::runtime_error("rank " + std::to_string(r) + ": " + errors[r]);
        if (stats) {
            *stats = mli_shard_stats{};
            for (int r = 0; r < n; ++r) {
                mli_engine_stats s{};
                if (mli_engine_get_stats(group->engines[r], &s) != 0) throw std::runtime_error(mli_engine_last_error());
                stats->total_tokens += s.total_tokens;
                stats->finished += s.finished;
            }
            stats->seconds = seconds;
            stats->iterations = iterations[0];
            stats->ranks_seen = group->ranks_seen;
            stats->gather_us = iterations[0] ? 1e6 * gather_seconds[0] / iterations[0] : 0.0;
        }
    })
}

int mli_shard_group_gathered(mli_shard_group* group, int rank, void** device_ptr, int* count) {
    if (!group || rank < 0 || rank >= group->n || !device_ptr || !count) { g_shard_error = "bad argument"; return -1; }
    *device_ptr = group->gathered[rank];
    *count = group->n * group->count;
    return 0;
}

mli_engine* mli_shard_group_engine(mli_shard_group* group, int rank) {
    if (!group || rank < 0 || rank >= group->n) return nullptr;
    return group->engines[rank];
}

}  // extern "C"
