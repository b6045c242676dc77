// multi.cpp -- several GPUs of one node behind the C-ABI (include/hafgrasp.h: haf_create_multi ...), for a C++ host such as
// the action server: ONE process, one engine + one host thread + one HIP stream per shard, one RCCL communicator over the
// distinct devices (ncclCommInitAll), collectives over xGMI.
//
// What is sharded is what the reference leaves independent: the rolls of a request (each roll re-bins the rotated cloud,
// server.cpp:376-385) and the clouds of a batch.  The only exchanges are
//   * ncclBroadcast of a device-resident cloud to the other GPUs (<= 6.3 MB; a host cloud goes to every GPU over its own PCIe
//     link instead),
//   * ONE ncclAllGather of the 16-byte roll records per request (rolls sharded): the all-gather form, not a max-reduce,
//     because the early exit with show_only_best_grasp makes the cross-roll rule order dependent (server.cpp:362-365,
//     953-960) -- every rank ends up with all n_rolls records and the sequential rule runs on them (haf_finalize),
//   * ONE ncclAllReduce(max, uint64) of a packed (vote, cloud) key per batch (clouds sharded) that elects the best grasp.
// Tens of bytes each: latency-bound, the 153 GB/s of an xGMI link do not matter here.
#include "../../include/hafgrasp.h"
#include "engine_internal.h"

#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

thread_local std::string g_multi_create_error;

// one persistent host thread per shard: the HIP calls of a shard always come from the same thread, with its device current
class Worker {
public:
    Worker() : th_([this] { loop(); }) {}
    ~Worker()
    {
        {
            std::lock_guard<std::mutex> l(m_);
            quit_ = true;
        }
        cv_.notify_all();
        th_.join();
    }
    void submit(std::function<void()> f)
    {
        {
            std::lock_guard<std::mutex> l(m_);
            job_ = std::move(f);
            busy_ = true;
        }
        cv_.notify_all();
    }
    void wait()
    {
        std::unique_lock<std::mutex> l(m_);
        done_.wait(l, [this] { return !busy_; });
    }

private:
    void loop()
    {
        for (;;) {
            std::function<void()> f;
            {
                std::unique_lock<std::mutex> l(m_);
                cv_.wait(l, [this] { return quit_ || job_; });
                if (quit_ && !job_) return;
                f = std::move(job_);
                job_ = nullptr;
            }
            try {
                f();
            } catch (...) {
                // jobs report through their own status word; nothing may escape a thread
            }
            {
                std::lock_guard<std::mutex> l(m_);
                busy_ = false;
            }
            done_.notify_all();
        }
    }
    std::mutex m_;
    std::condition_variable cv_, done_;
    std::function<void()> job_;
    bool busy_ = false, quit_ = false;
    std::thread th_;
};

struct Shard {
    int device = 0, rank = 0, slot = 0;       // slot: index among the shards of its rank
    haf_engine *eng = nullptr;
    Worker *worker = nullptr;
    int rc = 0;
    std::string err;
};

struct Rank {
    int device = 0;
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;             // collectives of this rank
    char *d_send = nullptr, *d_recv = nullptr;
    unsigned long long *d_key = nullptr;
    float *d_cloud = nullptr;                 // broadcast target for device-resident clouds
    size_t cloud_cap = 0;                     // floats
};

}  // namespace

struct haf_multi {
    haf_config cfg{};
    std::string feature_file, range_file, model_file;
    int mode = HAF_SHARD_ROLLS;
    std::vector<Shard> shards;
    std::vector<Rank> ranks;
    int shards_per_rank = 1;
    int block_records = 0;                    // roll records per shard in the all-gather (ceil(n_rolls / n_shards))
    std::vector<haf_roll_record> h_gather;    // rank 0's copy of the gathered records
    std::vector<std::vector<haf_roll_record>> h_all;   // every rank's copy (haf_multi_last_records)
    std::string error;
    int rccl_version = 0;
    // host wall-clock of the last sharded call's parts (haf_multi_last_timing): so that the first multi-GPU run explains itself
    float last_bcast_us = 0.0f, last_collective_us = 0.0f, last_total_ms = 0.0f;
    std::vector<float> last_shard_ms;
};

namespace {

int mfail(haf_multi *m, int code, const std::string &msg)
{
    m->error = msg;
    return code;
}

#define MHIP(m, call)                                                                          \
    do {                                                                                       \
        hipError_t err__ = (call);                                                             \
        if (err__ != hipSuccess) return mfail(m, HAF_E_DEVICE, std::string(#call) + ": " + hipGetErrorString(err__)); \
    } while (0)
#define MNCCL(m, call)                                                                         \
    do {                                                                                       \
        ncclResult_t err__ = (call);                                                           \
        if (err__ != ncclSuccess) return mfail(m, HAF_E_DEVICE, std::string(#call) + ": " + ncclGetErrorString(err__)); \
    } while (0)

// contiguous roll range of shard s: 36 rolls over 8 shards -> 5,5,5,5,4,4,4,4
void roll_range(int n_rolls, int n_shards, int s, int *first, int *count)
{
    const int q = n_rolls / n_shards, r = n_rolls % n_shards;
    *first = s * q + std::min(s, r);
    *count = q + (s < r ? 1 : 0);
}

// The partition haf_create_multi builds, without touching a device (also behind haf_multi_plan): rank and slot of every shard
// (ranks = distinct devices in order of first appearance), and for HAF_SHARD_ROLLS the roll range of every shard.
int plan_shards(const int32_t *devices, int n, int mode, int n_rolls, int *rank_of, int *slot_of, int *roll_first, int *roll_count,
                std::vector<int> *rank_dev, std::string *err)
{
    if (!devices || n < 1 || n > 64) { if (err) *err = "haf_create_multi: bad argument"; return HAF_E_ARG; }
    if (mode != HAF_SHARD_ROLLS && mode != HAF_SHARD_CLOUDS) { if (err) *err = "haf_create_multi: unknown shard mode"; return HAF_E_ARG; }
    if (mode == HAF_SHARD_ROLLS && n > n_rolls) { if (err) *err = "more shards than rolls"; return HAF_E_ARG; }
    std::vector<int> devs, per_rank;
    for (int s = 0; s < n; s++) {
        if (devices[s] < 0) { if (err) *err = "negative device ordinal in devices[]"; return HAF_E_ARG; }
        int rk = -1;
        for (size_t r = 0; r < devs.size(); r++) if (devs[r] == devices[s]) rk = (int)r;
        if (rk < 0) { devs.push_back(devices[s]); per_rank.push_back(0); rk = (int)devs.size() - 1; }
        if (rank_of) rank_of[s] = rk;
        if (slot_of) slot_of[s] = per_rank[(size_t)rk];
        per_rank[(size_t)rk]++;
        if (mode == HAF_SHARD_ROLLS && roll_first && roll_count) roll_range(n_rolls, n, s, &roll_first[s], &roll_count[s]);
    }
    for (int c : per_rank) if (c != per_rank[0]) { if (err) *err = "every device must appear the same number of times in devices[]"; return HAF_E_ARG; }
    if (rank_dev) *rank_dev = devs;
    return HAF_OK;
}

void run_on_all(haf_multi *m, const std::function<void(Shard &)> &f)
{
    for (Shard &sh : m->shards) sh.worker->submit([&sh, &f] { f(sh); });
    for (Shard &sh : m->shards) sh.worker->wait();
}

int first_error(haf_multi *m, const char *what)
{
    for (size_t s = 0; s < m->shards.size(); s++)
        if (m->shards[s].rc != HAF_OK) {
            char head[96];
            snprintf(head, sizeof head, "%s: shard %zu (device %d): ", what, s, m->shards[s].device);
            m->error = head + m->shards[s].err;
            return m->shards[s].rc;
        }
    return HAF_OK;
}

void destroy_multi(haf_multi *m)
{
    if (!m) return;
    for (Shard &sh : m->shards) {
        if (sh.worker && sh.eng) {
            haf_engine *e = sh.eng;
            sh.worker->submit([e] { haf_destroy(e); });
            sh.worker->wait();
        }
        delete sh.worker;
    }
    for (Rank &r : m->ranks) {
        (void)hipSetDevice(r.device);
        if (r.stream) (void)hipStreamSynchronize(r.stream);
        if (r.comm) (void)ncclCommDestroy(r.comm);
        if (r.d_send) (void)hipFree(r.d_send);
        if (r.d_recv) (void)hipFree(r.d_recv);
        if (r.d_key) (void)hipFree(r.d_key);
        if (r.d_cloud) (void)hipFree(r.d_cloud);
        if (r.stream) (void)hipStreamDestroy(r.stream);
    }
    delete m;
}

int create_multi(const haf_config *cfg, const int32_t *devices, int32_t n, int32_t mode, haf_multi **out)
{
    if (out) *out = nullptr;
    if (!cfg || !devices || !out || n < 1 || n > 64) { g_multi_create_error = "haf_create_multi: bad argument"; return HAF_E_ARG; }
    if (mode != HAF_SHARD_ROLLS && mode != HAF_SHARD_CLOUDS) { g_multi_create_error = "haf_create_multi: unknown shard mode"; return HAF_E_ARG; }
    if (!cfg->feature_file || !cfg->range_file || !cfg->model_file) { g_multi_create_error = "feature_file, range_file and model_file are required"; return HAF_E_ARG; }
    haf_multi *m = new haf_multi();
    struct Guard { haf_multi *m; ~Guard() { if (m) destroy_multi(m); } } guard{m};
    auto bail = [&](int code) { g_multi_create_error = m->error; return code; };
    m->cfg = *cfg;
    m->feature_file = cfg->feature_file; m->range_file = cfg->range_file; m->model_file = cfg->model_file;
    m->cfg.feature_file = m->feature_file.c_str(); m->cfg.range_file = m->range_file.c_str(); m->cfg.model_file = m->model_file.c_str();
    m->mode = mode;

    // ranks = distinct devices in order of first appearance; every rank must hold the same number of shards (equal all-gather blocks)
    m->shards.resize((size_t)n);
    std::vector<int> rank_of((size_t)n), slot_of((size_t)n), rank_dev;
    {
        std::string perr;
        const int prc = plan_shards(devices, n, mode, cfg->n_rolls, rank_of.data(), slot_of.data(), nullptr, nullptr, &rank_dev, &perr);
        if (prc != HAF_OK) { m->error = perr; return bail(prc); }
    }
    std::vector<int> per_rank(rank_dev.size(), 0);
    for (int d : rank_dev) { Rank r; r.device = d; m->ranks.push_back(r); }
    for (int s = 0; s < n; s++) {
        m->shards[(size_t)s].device = devices[s];
        m->shards[(size_t)s].rank = rank_of[(size_t)s];
        m->shards[(size_t)s].slot = slot_of[(size_t)s];
        per_rank[(size_t)rank_of[(size_t)s]]++;
    }
    m->shards_per_rank = per_rank[0];
    const int n_ranks = (int)m->ranks.size();
    m->block_records = (cfg->n_rolls + n - 1) / n;

    // engines: one per shard, created on the shard's own thread; capacity = the shard's share of the request
    haf_config ecfg = m->cfg;
    if (mode == HAF_SHARD_ROLLS) ecfg.max_rolls_per_call = m->block_records;
    else ecfg.max_clouds = (cfg->max_clouds + n - 1) / n;
    for (Shard &sh : m->shards) sh.worker = new Worker();
    run_on_all(m, [&](Shard &sh) {
        haf_config c = ecfg;
        c.device = sh.device;
        sh.rc = haf_create(&c, &sh.eng);
        if (sh.rc != HAF_OK) sh.err = haf_last_error(nullptr);
    });
    if (int rc = first_error(m, "haf_create")) return bail(rc);

    // one RCCL communicator per distinct device, all in this process
    std::vector<int> devs((size_t)n_ranks);
    std::vector<ncclComm_t> comms((size_t)n_ranks, nullptr);
    for (int r = 0; r < n_ranks; r++) devs[(size_t)r] = m->ranks[(size_t)r].device;
    {
        ncclResult_t rc = ncclCommInitAll(comms.data(), n_ranks, devs.data());
        if (rc != ncclSuccess) { m->error = std::string("ncclCommInitAll: ") + ncclGetErrorString(rc); return bail(HAF_E_DEVICE); }
    }
    (void)ncclGetVersion(&m->rccl_version);
    const size_t block_bytes = (size_t)m->block_records * sizeof(haf_roll_record) * (size_t)m->shards_per_rank;
    // every communicator belongs to its rank BEFORE anything below can fail: destroy_multi (the guard) then releases all of them
    for (int r = 0; r < n_ranks; r++) m->ranks[(size_t)r].comm = comms[(size_t)r];
    for (int r = 0; r < n_ranks; r++) {
        Rank &rk = m->ranks[(size_t)r];
        bool ok = hipSetDevice(rk.device) == hipSuccess;
        ok = ok && hipStreamCreateWithFlags(&rk.stream, hipStreamNonBlocking) == hipSuccess;
        ok = ok && hipMalloc((void **)&rk.d_send, std::max<size_t>(16, block_bytes)) == hipSuccess;
        ok = ok && hipMalloc((void **)&rk.d_recv, std::max<size_t>(16, block_bytes * (size_t)n_ranks)) == hipSuccess;
        ok = ok && hipMalloc((void **)&rk.d_key, sizeof(unsigned long long) * 2) == hipSuccess;
        if (!ok) { m->error = std::string("device buffers of rank ") + std::to_string(r) + ": " + hipGetErrorString(hipGetLastError()); return bail(HAF_E_DEVICE); }
    }
    m->h_all.assign((size_t)n_ranks, std::vector<haf_roll_record>());
    guard.m = nullptr;
    *out = m;
    return HAF_OK;
}

// A device-resident cloud lives on ONE GPU (that of shard 0): hand it to the others with one ncclBroadcast over xGMI.
// Returns per-rank pointers in dev_ptr.
int broadcast_cloud(haf_multi *m, const haf_cloud *cloud, std::vector<const float *> &dev_ptr)
{
    const int n_ranks = (int)m->ranks.size();
    dev_ptr.assign((size_t)n_ranks, cloud->xyz);
    if (n_ranks == 1) return HAF_OK;
    const size_t floats = cloud->n_points * cloud->stride_floats;
    for (int r = 0; r < n_ranks; r++) {
        Rank &rk = m->ranks[(size_t)r];
        if (rk.cloud_cap < floats) {
            MHIP(m, hipSetDevice(rk.device));
            if (rk.d_cloud) (void)hipFree(rk.d_cloud);
            rk.d_cloud = nullptr;
            MHIP(m, hipMalloc((void **)&rk.d_cloud, std::max<size_t>(floats, 1) * sizeof(float)));
            rk.cloud_cap = floats;
        }
    }
    // The broadcast runs on the ranks' private non-blocking streams: nothing orders it behind the stream that PRODUCED the cloud
    // on devices[0].  One device-wide synchronisation there (a few microseconds against a multi-megabyte broadcast) does.
    MHIP(m, hipSetDevice(m->ranks[0].device));
    MHIP(m, hipDeviceSynchronize());
    MNCCL(m, ncclGroupStart());
    for (int r = 0; r < n_ranks; r++) {
        Rank &rk = m->ranks[(size_t)r];
        ncclResult_t rc = ncclBroadcast(cloud->xyz, rk.d_cloud, floats, ncclFloat, 0, rk.comm, rk.stream);
        if (rc != ncclSuccess) { (void)ncclGroupEnd(); return mfail(m, HAF_E_DEVICE, std::string("ncclBroadcast: ") + ncclGetErrorString(rc)); }
    }
    MNCCL(m, ncclGroupEnd());
    for (int r = 0; r < n_ranks; r++) {
        MHIP(m, hipSetDevice(m->ranks[(size_t)r].device));
        MHIP(m, hipStreamSynchronize(m->ranks[(size_t)r].stream));
        dev_ptr[(size_t)r] = (r == 0) ? cloud->xyz : m->ranks[(size_t)r].d_cloud;
    }
    return HAF_OK;
}

int score_sharded(haf_multi *m, const haf_cloud *cloud, const haf_grasp_input *in, haf_grasp_output *out)
{
    if (!m) return HAF_E_ARG;
    if (!cloud || !in || !out) return mfail(m, HAF_E_ARG, "haf_score_sharded: null argument");
    if (m->mode != HAF_SHARD_ROLLS) return mfail(m, HAF_E_ARG, "haf_score_sharded needs a handle created with HAF_SHARD_ROLLS");
    const int n = (int)m->shards.size(), n_ranks = (int)m->ranks.size(), R = m->cfg.n_rolls;
    typedef std::chrono::steady_clock clk;
    auto us_since = [](clk::time_point t0) { return (float)std::chrono::duration<double, std::micro>(clk::now() - t0).count(); };
    const clk::time_point t_call = clk::now();
    m->last_bcast_us = 0.0f;
    m->last_shard_ms.assign((size_t)n, 0.0f);
    std::vector<const float *> dev_ptr;
    if (cloud->on_device == 1) {
        const clk::time_point t0 = clk::now();
        if (int rc = broadcast_cloud(m, cloud, dev_ptr)) return rc;
        m->last_bcast_us = us_since(t0);
    }
    const size_t rec_bytes = sizeof(haf_roll_record), block = (size_t)m->block_records * rec_bytes;

    // every shard scores its rolls and leaves the records in its rank's send block (device to device, on the engine's stream)
    run_on_all(m, [&](Shard &sh) {
        int first, count;
        roll_range(R, n, (int)(&sh - m->shards.data()), &first, &count);
        haf_cloud c = *cloud;
        if (cloud->on_device == 1) c.xyz = dev_ptr[(size_t)sh.rank];
        else c.on_device = 0;          // (a buffer registered with ONE engine is plain host memory to the shards' engines)
        std::vector<haf_roll_record> rec((size_t)count);
        const clk::time_point t0 = clk::now();
        sh.rc = haf_score_rolls(sh.eng, 1, &c, in, first, count, rec.data());
        m->last_shard_ms[(size_t)(&sh - m->shards.data())] = us_since(t0) * 1e-3f;
        if (sh.rc != HAF_OK) { sh.err = haf_last_error(sh.eng); return; }
        Rank &rk = m->ranks[(size_t)sh.rank];
        hipStream_t s = haf::engine_stream(sh.eng);
        char *dst = rk.d_send + (size_t)sh.slot * block;
        hipError_t e1 = hipMemsetAsync(dst, 0, block, s);
        hipError_t e2 = hipMemcpyAsync(dst, haf::engine_records_dev(sh.eng), (size_t)count * rec_bytes, hipMemcpyDeviceToDevice, s);
        hipError_t e3 = hipStreamSynchronize(s);
        if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) { sh.rc = HAF_E_DEVICE; sh.err = "staging the roll records for the all-gather failed"; }
    });
    if (int rc = first_error(m, "haf_score_sharded")) return rc;

    // ONE all-gather of the roll records: afterwards every rank holds the records of all n_rolls rolls
    const size_t send_bytes = block * (size_t)m->shards_per_rank;
    const clk::time_point t_gather = clk::now();
    MNCCL(m, ncclGroupStart());
    for (int r = 0; r < n_ranks; r++) {
        Rank &rk = m->ranks[(size_t)r];
        ncclResult_t rc = ncclAllGather(rk.d_send, rk.d_recv, send_bytes, ncclChar, rk.comm, rk.stream);
        if (rc != ncclSuccess) { (void)ncclGroupEnd(); return mfail(m, HAF_E_DEVICE, std::string("ncclAllGather: ") + ncclGetErrorString(rc)); }
    }
    MNCCL(m, ncclGroupEnd());
    std::vector<char> raw(send_bytes * (size_t)n_ranks);
    for (int r = 0; r < n_ranks; r++) {
        Rank &rk = m->ranks[(size_t)r];
        MHIP(m, hipSetDevice(rk.device));
        MHIP(m, hipMemcpyAsync(raw.data(), rk.d_recv, raw.size(), hipMemcpyDeviceToHost, rk.stream));
        MHIP(m, hipStreamSynchronize(rk.stream));
        // unpack: rank-major blocks, shard slots inside a rank's block -> roll order
        std::vector<haf_roll_record> &all = m->h_all[(size_t)r];
        all.assign((size_t)R, haf_roll_record());
        for (int s = 0; s < n; s++) {
            int first, count;
            roll_range(R, n, s, &first, &count);
            const Shard &sh = m->shards[(size_t)s];
            const char *src = raw.data() + (size_t)sh.rank * send_bytes + (size_t)sh.slot * block;
            memcpy(all.data() + first, src, (size_t)count * rec_bytes);
        }
    }
    m->last_collective_us = us_since(t_gather);       // the all-gather and every rank's copy of the gathered records to the host
    m->h_gather = m->h_all[0];
    // the sequential cross-roll rule and the pose on the full record set (what any rank could do now; rank 0 does)
    int rc = haf_finalize(m->shards[0].eng, in, m->h_gather.data(), out);
    if (rc != HAF_OK) return mfail(m, rc, haf_last_error(m->shards[0].eng));
    int64_t rechecked = 0;
    for (Shard &sh : m->shards) {
        int64_t a = 0, b = 0, c = 0;
        (void)haf_last_counts(sh.eng, &a, &b, &c);
        rechecked += b;
    }
    out->n_rechecked = rechecked;
    m->last_total_ms = us_since(t_call) * 1e-3f;
    return HAF_OK;
}

int score_batch_sharded(haf_multi *m, int32_t n_clouds, const haf_cloud *clouds, const haf_grasp_input *in, haf_grasp_output *out,
                        int32_t *best_cloud)
{
    if (!m) return HAF_E_ARG;
    if (!clouds || !in || !out || n_clouds < 1) return mfail(m, HAF_E_ARG, "haf_score_batch_sharded: null or empty argument");
    if (m->mode != HAF_SHARD_CLOUDS) return mfail(m, HAF_E_ARG, "haf_score_batch_sharded needs a handle created with HAF_SHARD_CLOUDS");
    if (n_clouds > m->cfg.max_clouds) return mfail(m, HAF_E_CAPACITY, "more clouds than max_clouds");
    const int n = (int)m->shards.size(), n_ranks = (int)m->ranks.size();
    for (int b = 0; b < n_clouds; b++)
        if (clouds[b].on_device == 1) return mfail(m, HAF_E_ARG, "haf_score_batch_sharded takes host clouds (each goes to its GPU over that GPU's PCIe link)");
    typedef std::chrono::steady_clock clk;
    auto us_since = [](clk::time_point t0) { return (float)std::chrono::duration<double, std::micro>(clk::now() - t0).count(); };
    const clk::time_point t_call = clk::now();
    m->last_bcast_us = 0.0f;
    m->last_shard_ms.assign((size_t)n, 0.0f);
    // cloud b -> shard b % n: independent requests, no data-path exchange at all
    std::vector<unsigned long long> shard_key((size_t)n, 0ull);
    run_on_all(m, [&](Shard &sh) {
        const int s = (int)(&sh - m->shards.data());
        std::vector<haf_cloud> cl;
        std::vector<haf_grasp_input> gi;
        std::vector<int> idx;
        for (int b = s; b < n_clouds; b += n) { cl.push_back(clouds[b]); cl.back().on_device = 0; gi.push_back(in[b]); idx.push_back(b); }
        sh.rc = HAF_OK;
        if (cl.empty()) return;
        std::vector<haf_grasp_output> o(cl.size());
        const clk::time_point t0 = clk::now();
        sh.rc = haf_score_batch(sh.eng, (int)cl.size(), cl.data(), gi.data(), o.data());
        m->last_shard_ms[(size_t)s] = us_since(t0) * 1e-3f;
        if (sh.rc != HAF_OK) { sh.err = haf_last_error(sh.eng); return; }
        unsigned long long key = 0;
        for (size_t k = 0; k < o.size(); k++) {
            out[idx[k]] = o[k];
            // larger vote wins, then the lower cloud index
            const unsigned long long kk = ((unsigned long long)(unsigned)(o[k].best_vote + 1000) << 32) | (unsigned)(0x7FFFFFFF - idx[k]);
            key = std::max(key, kk);
        }
        shard_key[(size_t)s] = key;
    });
    if (int rc = first_error(m, "haf_score_batch_sharded")) return rc;
    // ONE all-reduce(max) of the packed best-grasp key (north_star: "a single RCCL all-reduce of the best-grasp score")
    const clk::time_point t_red = clk::now();
    for (int r = 0; r < n_ranks; r++) {
        Rank &rk = m->ranks[(size_t)r];
        unsigned long long key = 0;
        for (int s = 0; s < n; s++) if (m->shards[(size_t)s].rank == r) key = std::max(key, shard_key[(size_t)s]);
        MHIP(m, hipSetDevice(rk.device));
        MHIP(m, hipMemcpyAsync(rk.d_key, &key, sizeof key, hipMemcpyHostToDevice, rk.stream));
        MHIP(m, hipStreamSynchronize(rk.stream));      // `key` is a stack variable
    }
    MNCCL(m, ncclGroupStart());
    for (int r = 0; r < n_ranks; r++) {
        Rank &rk = m->ranks[(size_t)r];
        ncclResult_t rc = ncclAllReduce(rk.d_key, rk.d_key + 1, 1, ncclUint64, ncclMax, rk.comm, rk.stream);
        if (rc != ncclSuccess) { (void)ncclGroupEnd(); return mfail(m, HAF_E_DEVICE, std::string("ncclAllReduce: ") + ncclGetErrorString(rc)); }
    }
    MNCCL(m, ncclGroupEnd());
    unsigned long long best = 0;
    for (int r = 0; r < n_ranks; r++) {
        Rank &rk = m->ranks[(size_t)r];
        unsigned long long k = 0;
        MHIP(m, hipSetDevice(rk.device));
        MHIP(m, hipMemcpyAsync(&k, rk.d_key + 1, sizeof k, hipMemcpyDeviceToHost, rk.stream));
        MHIP(m, hipStreamSynchronize(rk.stream));
        if (r == 0) best = k;
        else if (k != best) return mfail(m, HAF_E_INTERNAL, "ranks disagree on the all-reduced best-grasp key");
    }
    if (best_cloud) *best_cloud = (int32_t)(0x7FFFFFFF - (unsigned)(best & 0xFFFFFFFFu));
    m->last_collective_us = us_since(t_red);
    m->last_total_ms = us_since(t_call) * 1e-3f;
    return HAF_OK;
}

template <class F> int guarded(std::string *err, F &&f)
{
    try {
        return f();
    } catch (const std::bad_alloc &) {
        if (err) *err = "out of host memory";
    } catch (const std::exception &ex) {
        if (err) *err = std::string("internal error: ") + ex.what();
    } catch (...) {
        if (err) *err = "internal error (unknown exception)";
    }
    return HAF_E_INTERNAL;
}

}  // namespace

// The entry points below walk over the ranks with hipSetDevice on the CALLER's thread: put its current device back on the way
// out, so that a C++ or torch host is not left on the last rank's GPU.
struct DeviceGuard {
    int dev = -1;
    DeviceGuard() { if (hipGetDevice(&dev) != hipSuccess) dev = -1; }
    ~DeviceGuard() { if (dev >= 0) (void)hipSetDevice(dev); }
};

extern "C" {

int haf_create_multi(const haf_config *cfg, const int32_t *devices, int32_t n_devices, int32_t shard_mode, haf_multi **out)
{
    DeviceGuard keep;
    return guarded(&g_multi_create_error, [&] { return create_multi(cfg, devices, n_devices, shard_mode, out); });
}

void haf_destroy_multi(haf_multi *m) { DeviceGuard keep; destroy_multi(m); }

const char *haf_multi_last_error(const haf_multi *m) { return m ? m->error.c_str() : g_multi_create_error.c_str(); }

int haf_score_sharded(haf_multi *m, const haf_cloud *cloud, const haf_grasp_input *in, haf_grasp_output *out)
{
    DeviceGuard keep;
    return guarded(m ? &m->error : nullptr, [&] { return score_sharded(m, cloud, in, out); });
}

int haf_score_batch_sharded(haf_multi *m, int32_t n_clouds, const haf_cloud *clouds, const haf_grasp_input *in, haf_grasp_output *out,
                            int32_t *best_cloud)
{
    DeviceGuard keep;
    return guarded(m ? &m->error : nullptr, [&] { return score_batch_sharded(m, n_clouds, clouds, in, out, best_cloud); });
}

int haf_multi_plan(const int32_t *devices, int32_t n_devices, int32_t shard_mode, int32_t n_rolls, int32_t *rank_of, int32_t *slot_of,
                   int32_t *roll_first, int32_t *roll_count, int32_t *n_ranks)
{
    std::vector<int> devs;
    std::string err;
    const int rc = plan_shards(devices, n_devices, shard_mode, n_rolls, rank_of, slot_of, roll_first, roll_count, &devs, &err);
    if (rc != HAF_OK) { g_multi_create_error = err; return rc; }
    if (n_ranks) *n_ranks = (int32_t)devs.size();
    return HAF_OK;
}

int haf_multi_info(const haf_multi *m, int32_t *n_shards, int32_t *n_ranks, int32_t *rccl_version)
{
    if (!m) return HAF_E_ARG;
    if (n_shards) *n_shards = (int32_t)m->shards.size();
    if (n_ranks) *n_ranks = (int32_t)m->ranks.size();
    if (rccl_version) *rccl_version = m->rccl_version;
    return HAF_OK;
}

haf_engine *haf_multi_engine(haf_multi *m, int32_t shard)
{
    if (!m || shard < 0 || shard >= (int32_t)m->shards.size()) return nullptr;
    return m->shards[(size_t)shard].eng;
}

int haf_multi_last_timing(const haf_multi *m, float *total_ms, float *bcast_us, float *collective_us, float *shard_ms)
{
    if (!m) return HAF_E_ARG;
    if (total_ms) *total_ms = m->last_total_ms;
    if (bcast_us) *bcast_us = m->last_bcast_us;
    if (collective_us) *collective_us = m->last_collective_us;
    if (shard_ms) for (size_t s = 0; s < m->shards.size(); s++) shard_ms[s] = s < m->last_shard_ms.size() ? m->last_shard_ms[s] : 0.0f;
    return HAF_OK;
}

int haf_multi_last_records(const haf_multi *m, int32_t rank, haf_roll_record *records)
{
    if (!m || !records || rank < 0 || rank >= (int32_t)m->h_all.size()) return HAF_E_ARG;
    if (m->h_all[(size_t)rank].size() != (size_t)m->cfg.n_rolls) return HAF_E_ARG;
    memcpy(records, m->h_all[(size_t)rank].data(), (size_t)m->cfg.n_rolls * sizeof(haf_roll_record));
    return HAF_OK;
}

}  // extern "C"
