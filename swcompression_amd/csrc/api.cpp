// api.cpp -- C ABI of libswc_hip.so: device management, the batched launch, and the host-side batch
// runner used by every single-shot entry point.  Compiled with hipcc.  There is deliberately no CPU
// decode path in this library: without a usable gfx950 device every decode entry point returns
// SWC_E_DEVICE.
#include <algorithm>
#include <atomic>
#include <map>
#include <chrono>
#include <functional>
#include <stdio.h>
#include <stdlib.h>
#include <mutex>
#include <thread>
#include <system_error>
#include <vector>
#include <string.h>
#include "host_util.h"
#include "launch.h"

static_assert(sizeof(swc_job) == sizeof(swc::Job), "swc_job and swc::Job must have the same layout");
static_assert(offsetof(swc_job, status) == offsetof(swc::Job, status), "layout");
static_assert(offsetof(swc_job, dict_len) == offsetof(swc::Job, dict_len), "layout");

namespace swc {

// What the library keeps between calls so that the next call need not ask the driver / the kernel for it again: freed device
// memory in the device's pool, the calling thread's page-locked staging buffers, the large host results handed back through
// swc_free().  All three are bounded, the bounds can be set (swc_set_tuning "pool_keep_mib" / "pinned_keep_mib" /
// "result_cache_mib", or the environment variables SWC_POOL_KEEP_MIB / SWC_PINNED_KEEP_MIB / SWC_RESULT_CACHE_MIB read once),
// and swc_trim() gives everything back (a process with N threads held 2 N GiB of pinned memory and 8 GiB of HBM).
static size_t env_mib(const char* name, size_t dflt) {
    const char* v = getenv(name);
    if (!v || !*v) return dflt;
    char* end = nullptr;
    const unsigned long long x = strtoull(v, &end, 10);
    return end && *end == 0 && x <= (1ull << 20) ? (size_t)x : dflt;
}
static std::atomic<size_t> g_pool_keep{env_mib("SWC_POOL_KEEP_MIB", 4096) << 20};       // a BGZF file of 4,096 members needs a workspace of 1.1 GB per call, an xz file of 512 blocks one of 3 GB
static std::atomic<size_t> g_pinned_keep{env_mib("SWC_PINNED_KEEP_MIB", 512) << 20};    // per thread and direction
static std::atomic<size_t> g_result_cache{env_mib("SWC_RESULT_CACHE_MIB", 512) << 20};  // process-wide

static std::once_flag g_dev_once;
static bool g_dev_ok = false;
static std::mutex g_dev_mu;
static std::vector<int> g_dev_state;   // per device: 0 not looked at, 1 gfx950, 2 something else

// The code object in this library is gfx950 only.
bool device_is_gfx950(int dev) {
    std::lock_guard<std::mutex> lk(g_dev_mu);
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { (void)hipGetLastError(); return false; }
    if (dev < 0 || dev >= n) return false;
    if (g_dev_state.size() < (size_t)n) g_dev_state.resize(n, 0);
    if (g_dev_state[dev] == 0) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) { (void)hipGetLastError(); return false; }
        g_dev_state[dev] = strncmp(prop.gcnArchName, "gfx950", 6) == 0 ? 1 : 2;
        if (g_dev_state[dev] == 1) {   // keep freed staging buffers of the single-shot calls in the pool instead of returning them to the OS
            hipMemPool_t pool;
            uint64_t keep = g_pool_keep.load();
            if (hipDeviceGetDefaultMemPool(&pool, dev) == hipSuccess) (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
            (void)hipGetLastError();
        }
    }
    return g_dev_state[dev] == 1;
}

bool device_ready() {
    std::call_once(g_dev_once, [] {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return; }
        g_dev_ok = device_is_gfx950(dev);
    });
    return g_dev_ok;
}

static hipError_t launch_codec(int codec, Job* jobs, size_t n, void* ws, size_t ws_bytes, hipStream_t stream, const uint64_t* ws_off = nullptr) {
    switch (codec) {
        case SWC_CODEC_DEFLATE: return launch_inflate(jobs, n, ws, ws_bytes, stream, ws_off);
        case SWC_CODEC_LZ4_BLOCK: return launch_lz4(jobs, n, ws, ws_bytes, stream, ws_off);
        case SWC_CODEC_LZMA2: return launch_lzma(true, jobs, n, ws_bytes >= n * lzma_spill_bytes_per_job() ? ws : nullptr, stream);
        case SWC_CODEC_LZMA: return launch_lzma(false, jobs, n, ws_bytes >= n * lzma_spill_bytes_per_job() ? ws : nullptr, stream);
        case SWC_CODEC_BZIP2_BLOCK: return launch_bzip2(jobs, n, ws, ws_bytes, stream);
        case SWC_CODEC_DELTA: return launch_delta(jobs, n, stream);
        case SWC_CODEC_LZ4_COMPRESS: return launch_lz4_compress(jobs, n, stream);
        case SWC_CODEC_DEFLATE_COMPRESS: return launch_deflate_compress(jobs, n, stream);
        default: return hipErrorInvalidValue;
    }
}

static size_t default_cap(int codec, const HostUnit& u) {
    // A declared size is only a starting point (units that report SWC_E_CAPACITY are relaunched with more): a damaged
    // size field must not turn into a terabyte allocation and an SWC_E_DEVICE for the whole batch.
    if (u.cap_hint) return u.cap_exact ? u.cap_hint : std::min<size_t>(u.cap_hint, std::max<size_t>((size_t)64 << 20, u.in_len * 2048));
    switch (codec) {
        case SWC_CODEC_DEFLATE: return std::max<size_t>(65536, u.in_len * 4 + 1024);
        case SWC_CODEC_LZ4_BLOCK: return std::max<size_t>(65536, u.in_len * 4 + 1024);
        default: return std::max<size_t>(1 << 20, u.in_len * 8);
    }
}

// ---- host results: large ones are recycled ---------------------------------------------------------------------------------
// swc_free() of a result of kResultCacheMin bytes and more parks the buffer (up to kResultCacheMax bytes in all, the largest
// first to go); the next result of about that size -- at most twice as small -- takes it over with its pages mapped.
// Everything else is plain malloc / free.
constexpr size_t kResultCacheMin = (size_t)4 << 20;
static std::mutex g_res_mu;
static std::map<void*, size_t> g_res_live;                 // large results in the hands of callers: pointer -> capacity
static std::vector<std::pair<void*, size_t>> g_res_parked;  // freed ones
uint8_t* host_result(size_t n) {
    if (n >= kResultCacheMin) {
        std::lock_guard<std::mutex> lk(g_res_mu);
        size_t best = g_res_parked.size();
        for (size_t i = 0; i < g_res_parked.size(); i++)
            if (g_res_parked[i].second >= n && g_res_parked[i].second <= 2 * n && (best == g_res_parked.size() || g_res_parked[i].second < g_res_parked[best].second)) best = i;
        if (best != g_res_parked.size()) {
            auto e = g_res_parked[best];
            g_res_parked.erase(g_res_parked.begin() + (long)best);
            g_res_live[e.first] = e.second;
            return static_cast<uint8_t*>(e.first);
        }
        void* p = malloc(n);
        if (p) g_res_live[p] = n;
        return static_cast<uint8_t*>(p);
    }
    return static_cast<uint8_t*>(malloc(n ? n : 1));
}
void host_result_free(void* p) {
    if (!p) return;
    {
        std::lock_guard<std::mutex> lk(g_res_mu);
        auto it = g_res_live.find(p);
        if (it != g_res_live.end()) {
            const size_t cap = it->second;
            g_res_live.erase(it);
            size_t held = cap;
            for (auto& e : g_res_parked) held += e.second;
            g_res_parked.emplace_back(p, cap);
            while (held > g_result_cache.load() && !g_res_parked.empty()) {   // over the limit: the largest go back to the system
                size_t big = 0;
                for (size_t i = 1; i < g_res_parked.size(); i++) if (g_res_parked[i].second > g_res_parked[big].second) big = i;
                held -= g_res_parked[big].second;
                free(g_res_parked[big].first);
                g_res_parked.erase(g_res_parked.begin() + (long)big);
            }
            return;
        }
    }
    free(p);
}

static std::atomic<long long> g_stats[3];
void stat_add(int which, long long v) { if (which >= 0 && which < 3) g_stats[which] += v; }

// What a thread keeps page-locked between calls, per direction.  Pinning is the expensive part of a large single-shot call
// (a 268 MB result: tens of milliseconds to lock against 5 ms to copy), so a buffer that served an archive of a gigabyte
// stays for the next one; only a buffer beyond that is released when its call is over.

// Two page-locked staging buffers per calling thread (0: host -> device, 1: device -> host), grown on demand, released
// when the thread ends.
uint8_t* pinned_stage(int which, size_t n) {
    struct Buf {
        void* p = nullptr;
        size_t cap = 0;
        ~Buf() { if (p) (void)hipHostFree(p); }
    };
    static thread_local Buf bufs[2];
    Buf& b = bufs[which & 1];
    if (which & 6) {   // trim request (2: pinned_trim after a call -- what is over the limit; 4: swc_trim -- everything)
        if (b.p && ((which & 4) || b.cap > g_pinned_keep.load())) { (void)hipHostFree(b.p); b.p = nullptr; b.cap = 0; }
        return nullptr;
    }
    if (b.cap < n) {
        if (b.p) { (void)hipHostFree(b.p); b.p = nullptr; b.cap = 0; }
        size_t want = std::max<size_t>(n + n / 4, (size_t)1 << 20);
        if (hipHostMalloc(&b.p, want, hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); b.p = nullptr; return nullptr; }
        b.cap = want;
    }
    return static_cast<uint8_t*>(b.p);
}
// A thread that staged a multi-gigabyte container once would hold that much page-locked memory until it ends: buffers above
// the limit ("pinned_keep_mib") are released when the call that needed them is over (round-2 advisor; round 4: the limit was 256 MiB, and a
// 268 MB BGZF file pinned its buffers again on every call).
static void pinned_trim() {
    for (int w = 0; w < 2; w++) (void)pinned_stage(w | 2, 0);
}

// The outputs of a launch leave the pinned staging buffer for where they are wanted (HostUnit::dst, else HostUnit::out): by
// several threads when there is a lot to move (one core copies 10 GB/s; PCIe delivered it at 50).
struct CopyOut { HostUnit* u; const uint8_t* src; size_t n; };
static void copy_range(std::vector<CopyOut>& v, size_t lo, size_t hi) {
    for (size_t i = lo; i < hi; i++) {
        HostUnit& u = *v[i].u;
        if (u.in_dst) { if (v[i].n) memcpy(u.dst, v[i].src, v[i].n); u.out.clear(); }
        else u.out.assign(v[i].src, v[i].src + v[i].n);
    }
}
static void copy_out(std::vector<CopyOut>& v) {
    size_t total = 0;
    for (const CopyOut& c : v) total += c.n;
    unsigned hw = std::thread::hardware_concurrency();
    size_t nt = std::min<size_t>({(size_t)8, hw ? (size_t)hw : (size_t)1, total >> 24, v.size()});   // a thread per 16 MB, eight at most
    if (nt < 2) { copy_range(v, 0, v.size()); return; }
    std::vector<std::thread> th;
    size_t i = 0, acc = 0, done = 0;   // done: entries a thread has taken
    try {
        for (size_t t = 0; t < nt; t++) {   // contiguous ranges of about total / nt bytes
            const size_t lo = i, target = total / nt * (t + 1);
            while (i < v.size() && (i == lo || acc + v[i].n <= target)) acc += v[i++].n;
            if (t + 1 == nt) i = v.size();
            th.emplace_back(copy_range, std::ref(v), lo, i);
            done = i;
        }
    } catch (const std::system_error&) {}   // no more threads to be had: this one copies what nobody took (never std::terminate)
    if (done < v.size()) copy_range(v, done, v.size());
    for (auto& x : th) x.join();
}

static int run_units_impl(int codec, std::vector<HostUnit>& units);
int run_units(int codec, std::vector<HostUnit>& units) {
    const int st = run_units_impl(codec, units);
    pinned_trim();
    return st;
}
static int run_units_impl(int codec, std::vector<HostUnit>& units) {
    if (!device_ready()) return SWC_E_DEVICE;
    const size_t n = units.size();
    if (n == 0) return SWC_OK;
    hipStream_t stream = hipStreamPerThread;   // every calling thread stages, launches and waits on its own stream
    Trace tr;
    std::vector<size_t> pending(n);
    for (size_t i = 0; i < n; i++) pending[i] = i;
    std::vector<size_t> cap(n);
    for (size_t i = 0; i < n; i++) cap[i] = default_cap(codec, units[i]);
    std::vector<size_t> ws_extra(n, 0);   // Deflate / LZ4: record room beyond what the capacity implies (a unit that reported SWC_E_NEED_WORKSPACE)

    bool want_ws = codec == SWC_CODEC_BZIP2_BLOCK || codec == SWC_CODEC_DEFLATE || codec == SWC_CODEC_LZ4_BLOCK;
    // LZMA / LZMA2: without a workspace every literal coder of a stream sits in LDS (5 streams per CU); with one, LDS caches four
    // LINES (a third of a coder each) and the coders live in the workspace (32 streams per CU, lzma_wave.h).  A batch that more than fills the 5-stream layout takes the workspace from the
    // start -- the container paths (xz, 7z, .lzma, unarchive_many) then run the kernel bench.py measures; a small
    // batch gets one only after a unit reported SWC_E_NEED_WORKSPACE (lc + lp > 4).
    // The up-front workspace is OPPORTUNISTIC: if the device cannot give it -- or only for fewer streams than the
    // 5-stream layout holds anyway -- the batch runs without one, as it did before, and only units that report
    // SWC_E_NEED_WORKSPACE come back for it.
    bool opportunistic_ws = false;
    if ((codec == SWC_CODEC_LZMA || codec == SWC_CODEC_LZMA2) && n > 5u * 256u) { want_ws = true; opportunistic_ws = true; }
    const bool per_job_ws = codec == SWC_CODEC_DEFLATE || codec == SWC_CODEC_LZ4_BLOCK;   // areas sized from each unit's own capacity
    // rounds: 12 relaunches for growing capacities, plus the extra rounds that splitting by workspace size takes
    for (int round = 0, grow_rounds = 0; grow_rounds < 12 && round < 256 && !pending.empty(); round++) {
        // Codecs whose workspace areas are one size for the whole launch (bzip2: 9 x the capacity): ONE big unit among many
        // small ones must not size everybody's area.  The units are taken in order of capacity as long as count x area stays
        // within a budget; the others wait for the next round.
        std::vector<size_t> deferred;
        if (want_ws && !per_job_ws && pending.size() > 1) {
            std::stable_sort(pending.begin(), pending.end(), [&](size_t a, size_t b) { return cap[a] < cap[b]; });
            // (what the device has free right now, not a constant: other threads and other entries of a device list launch too)
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); free_b = (size_t)16 << 30; }
            size_t io_bytes = 0;   // what the inputs and outputs of the launch take themselves
            for (size_t i : pending) io_bytes += units[i].in_len + cap[i] + 64;
            const size_t budget = std::min<size_t>((size_t)48 << 30, (free_b > io_bytes ? free_b - io_bytes : 0) / 2);
            size_t take = 1;
            while (take < pending.size() && swc_batch_workspace_bytes(codec, take + 1, cap[pending[take]]) <= budget) take++;
            if (opportunistic_ws && take < pending.size() && take < 5u * 256u) {   // not worth it: everybody at once, without
                want_ws = false;
                opportunistic_ws = false;
                take = pending.size();
            }
            deferred.assign(pending.begin() + take, pending.end());
            pending.resize(take);
        }
        if (deferred.empty()) grow_rounds++;
        const size_t m = pending.size();
        std::vector<size_t> in_off(m), out_off(m), dict_off(m);
        // Host buffers are staged once: units that share a buffer (the bzip2 blocks of a stream) or that are sub-ranges of
        // one `base` buffer (the entries of a container, the members of a file) address the one staged copy.
        struct Staged { const uint8_t* p; size_t len, off; };
        std::vector<Staged> staged;
        std::map<std::pair<const uint8_t*, size_t>, size_t> seen;
        size_t in_total = 0, out_total = 0;
        for (size_t k = 0; k < m; k++) {
            const HostUnit& u = units[pending[k]];
            const bool sub = u.base && u.in >= u.base && u.in + u.in_len <= u.base + u.base_len;
            const uint8_t* b = sub ? u.base : u.in;
            const size_t bl = sub ? u.base_len : u.in_len;
            auto key = std::make_pair(b, bl);
            auto it = seen.find(key);
            size_t at;
            if (it != seen.end()) at = it->second;
            else { at = in_total; seen[key] = at; staged.push_back(Staged{b, bl, at}); in_total += (bl + 15) & ~(size_t)15; }
            in_off[k] = at + (sub ? (size_t)(u.in - u.base) : 0);
            dict_off[k] = in_total;
            if (u.dict) in_total += (u.dict_len + 15) & ~(size_t)15;
            out_off[k] = out_total;
            out_total += (cap[pending[k]] + 15) & ~(size_t)15;
        }
        // workspace: per-job areas (prefix-summed) where the codec sizes them from the capacity, else one size for all
        std::vector<uint64_t> ws_off;
        size_t ws_bytes = 0;
        if (want_ws && per_job_ws) {
            ws_off.resize(m + 1);
            for (size_t k = 0; k < m; k++) { ws_off[k] = ws_bytes; ws_bytes += inflate_ws_bytes_per_job(cap[pending[k]]) + ws_extra[pending[k]]; }
            ws_off[m] = ws_bytes;
        } else if (want_ws) {
            size_t mx = 0;
            for (size_t k = 0; k < m; k++) mx = std::max(mx, cap[pending[k]]);
            ws_bytes = swc_batch_workspace_bytes(codec, m, mx);
        }
        // ONE device buffer [inputs | job records | workspace offsets | checksums | outputs] and one more for the workspace: one
        // copy up (everything in front of the checksums), one copy down (everything behind the inputs) -- a single-shot call
        // is a handful of driver calls, and each costs as much as the decode of a 64 KiB block
        int sum_kind = 0;
        for (size_t k = 0; k < m && !sum_kind; k++) sum_kind = units[pending[k]].sum_kind;
        const size_t jobs_bytes = (m * sizeof(Job) + 15) & ~(size_t)15, off_bytes = (ws_off.size() * sizeof(uint64_t) + 15) & ~(size_t)15;
        const size_t sum_bytes = sum_kind ? (m * sizeof(uint64_t) + 15) & ~(size_t)15 : 0;
        const size_t in_bytes = in_total + 16;
        const size_t up_bytes = in_bytes + jobs_bytes + off_bytes;                 // host -> device
        const size_t down_bytes = jobs_bytes + off_bytes + sum_bytes + out_total;   // device -> host
        tr.mark("plan");
        DevBuf d_all(in_bytes + down_bytes + 16), d_ws(ws_bytes);
        tr.mark("device buffers", in_bytes + down_bytes + ws_bytes);
        if (d_all.ok() && !d_ws.ok() && opportunistic_ws) {   // no room for the optional workspace: without it
            want_ws = false;
            opportunistic_ws = false;
            pending.insert(pending.end(), deferred.begin(), deferred.end());
            round--;
            continue;
        }
        if (!d_all.ok() || !d_ws.ok()) return SWC_E_DEVICE;
        uint8_t* const d_in = d_all.u8();
        uint8_t* const d_jobs = d_in + in_bytes;
        uint8_t* const d_sums = d_jobs + jobs_bytes + off_bytes;
        uint8_t* const d_out = d_sums + sum_bytes;

        std::vector<uint8_t> up_fallback;
        uint8_t* up = pinned_stage(0, up_bytes);
        if (!up) { up_fallback.resize(up_bytes); up = up_fallback.data(); }
        for (const Staged& st : staged) if (st.len) memcpy(up + st.off, st.p, st.len);
        Job* jobs = reinterpret_cast<Job*>(up + in_bytes);
        for (size_t k = 0; k < m; k++) {
            const HostUnit& u = units[pending[k]];
            if (u.dict && u.dict_len) memcpy(up + dict_off[k], u.dict, u.dict_len);
            Job& j = jobs[k];
            j.in = d_in + in_off[k];
            j.in_len = u.in_len;
            j.out = d_out + out_off[k];
            j.out_cap = cap[pending[k]];
            j.out_len = 0;
            j.in_consumed = 0;
            j.status = SWC_E_DEVICE;
            j.aux = u.aux;
            j.dict = u.dict ? d_in + dict_off[k] : reinterpret_cast<const uint8_t*>((uintptr_t)u.dict_value);
            j.dict_len = u.dict ? u.dict_len : u.extra;
        }
        if (!ws_off.empty()) memcpy(up + in_bytes + jobs_bytes, ws_off.data(), ws_off.size() * sizeof(uint64_t));
        tr.mark("stage inputs (pinned)", up_bytes);
        if (hipMemcpyAsync(d_in, up, up_bytes, hipMemcpyHostToDevice, stream) != hipSuccess) return SWC_E_DEVICE;
        const uint64_t* d_off = !ws_off.empty() ? reinterpret_cast<const uint64_t*>(d_jobs + jobs_bytes) : nullptr;
        if (launch_codec(codec, reinterpret_cast<Job*>(d_jobs), m, d_ws.ptr(), ws_bytes, stream, d_off) != hipSuccess) return SWC_E_DEVICE;
        if (sum_kind && launch_checksum(sum_kind, reinterpret_cast<const Job*>(d_jobs), m, reinterpret_cast<uint64_t*>(d_sums), stream) != hipSuccess) return SWC_E_DEVICE;
        stat_add(0, 1);
        stat_add(1, (long long)m);
        // device -> host: the job records (they say how much of every output exists), the checksums, the outputs
        std::vector<uint8_t> down_fallback;
        uint8_t* down = pinned_stage(1, down_bytes);
        if (!down) { down_fallback.resize(down_bytes); down = down_fallback.data(); }
        if (hipMemcpyAsync(down, d_jobs, down_bytes, hipMemcpyDeviceToHost, stream) != hipSuccess) return SWC_E_DEVICE;
        tr.mark("issue launch + copies");
        if (hipStreamSynchronize(stream) != hipSuccess) return SWC_E_DEVICE;
        tr.mark("H2D + kernels + D2H", down_bytes);
        const Job* res = reinterpret_cast<const Job*>(down);
        const uint64_t* sums = reinterpret_cast<const uint64_t*>(down + jobs_bytes + off_bytes);
        const uint8_t* outs = down + jobs_bytes + off_bytes + sum_bytes;

        std::vector<size_t> next;
        std::vector<CopyOut> done;
        done.reserve(m);
        for (size_t k = 0; k < m; k++) {
            HostUnit& u = units[pending[k]];
            const Job& j = res[k];
            if (j.status == SWC_E_NEED_WORKSPACE && !want_ws) {
                next.push_back(pending[k]);
                continue;
            }
            if (j.status == SWC_E_NEED_WORKSPACE && per_job_ws && ws_extra[pending[k]] == 0) {
                // The record list is sized from the capacity (one record per three output bytes and some); a stream can need
                // more -- many tiny blocks, matches of the shortest codes.  Once more with room for a record per two input
                // bits, which no stream exceeds (round-2 advisor: the internal status must not escape).
                ws_extra[pending[k]] = ((u.in_len * 16 + ((size_t)64 << 10)) + 15) & ~(size_t)15;
                next.push_back(pending[k]);
                continue;
            }
            if (j.status == SWC_E_NEED_WORKSPACE && codec == SWC_CODEC_BZIP2_BLOCK && cap[pending[k]] < ((size_t)15 << 20)) {
                cap[pending[k]] = std::min<size_t>(cap[pending[k]] * 4, (size_t)15 << 20);  // L outgrew the workspace sized from cap
                next.push_back(pending[k]);
                continue;
            }
            // (bzip2: a block of the largest BWT column the engine takes expands to at most 52 x 16,000,000 bytes under RLE1)
            const size_t grow_limit = codec == SWC_CODEC_BZIP2_BLOCK ? (size_t)1 << 30 : (size_t)1 << 34;
            if (j.status == SWC_E_CAPACITY && !u.cap_exact && cap[pending[k]] < grow_limit) {
                size_t want = j.out_len > cap[pending[k]] ? (size_t)j.out_len : cap[pending[k]] * 4;
                cap[pending[k]] = want;
                next.push_back(pending[k]);
                continue;
            }
            // a bzip2 block whose BWT column outgrows the largest workspace (16,000,000 bytes, 17 x the largest block an
            // encoder writes; the reference enforces no block size, App. A B5) is beyond the engine's capacity -- a
            // documented status, not the internal "needs a workspace"
            u.status = (j.status == SWC_E_NEED_WORKSPACE && codec == SWC_CODEC_BZIP2_BLOCK) ? (int)SWC_E_CAPACITY : j.status;
            u.aux_out = j.aux;
            u.in_consumed = (size_t)j.in_consumed;
            const size_t produced = (size_t)std::min<uint64_t>(j.out_len, j.out_cap);
            u.out_size = produced;
            u.in_dst = u.dst != nullptr && produced <= u.dst_cap;
            if (sum_kind && u.sum_kind == sum_kind) { u.sum = sums[k]; u.sum_valid = true; }
            done.push_back(CopyOut{&u, outs + out_off[k], produced});
        }
        copy_out(done);
        tr.mark("copy out", out_total);
        if (!next.empty() && !want_ws)
            for (size_t k = 0; k < m; k++)
                if (res[k].status == SWC_E_NEED_WORKSPACE) { want_ws = true; break; }
        next.insert(next.end(), deferred.begin(), deferred.end());
        pending.swap(next);
    }
    for (size_t i : pending) units[i].status = SWC_E_CAPACITY;
    return SWC_OK;
}

}  // namespace swc

using namespace swc;

extern "C" {

int swc_set_tuning(const char* key, int value) try {
    if (!key) return SWC_E_INVALID_ARGUMENT;
    if (!strcmp(key, "phase_timing") && (value == 0 || value == 1)) { set_phase_timing(value); return SWC_OK; }
    if (!strcmp(key, "lzma_coder_cache") && (value == 0 || value == 1)) { set_lzma_coder_cache(value); return SWC_OK; }
    if (!strcmp(key, "lz_copier") && value >= -2 && value <= 2) { set_lz_copier(value); return SWC_OK; }
    if (!strcmp(key, "deflate_team") && value >= -1 && value <= 1) { set_deflate_team(value); return SWC_OK; }
    if (!strcmp(key, "bzip2_hot_cxx") && (value == 0 || value == 1)) { set_bzip2_hot_cxx(value); return SWC_OK; }
    if (!strcmp(key, "bzip2_team_walk") && value >= 0 && value <= 2) { set_bzip2_team_walk(value); return SWC_OK; }
    if (!strcmp(key, "bzip2_team_per_cu") && value >= 1 && value <= 2) { set_bzip2_team_per_cu(value); return SWC_OK; }
    if (!strcmp(key, "pinned_keep_mib") && value >= 0) { g_pinned_keep = (size_t)value << 20; return SWC_OK; }
    if (!strcmp(key, "result_cache_mib") && value >= 0) { g_result_cache = (size_t)value << 20; return SWC_OK; }
    if (!strcmp(key, "pool_keep_mib") && value >= 0) {   // applies to the current device at once, to the others when they are first used
        g_pool_keep = (size_t)value << 20;
        int dev = 0;
        hipMemPool_t pool;
        uint64_t keep = g_pool_keep.load();
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetDefaultMemPool(&pool, dev) == hipSuccess) (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
        (void)hipGetLastError();
        return SWC_OK;
    }
    return SWC_E_INVALID_ARGUMENT;
} catch (...) {   // std::bad_alloc / length_error from a size taken from the input: never through the C boundary
    return SWC_E_DEVICE;
}

// Profile builds (-DSWC_PROFILE) only: a device buffer of 32 x u64 per job that the Deflate kernels fill with cycle counts.
int swc_set_profile_buffer(void* device_ptr) { set_profile_buffer(device_ptr); return SWC_OK; }

long long swc_stat(const char* key) {
    if (!key) return -1;
    if (!strcmp(key, "launches")) return g_stats[0].load();
    if (!strcmp(key, "units")) return g_stats[1].load();
    if (!strcmp(key, "xz_cache_hits")) return g_stats[2].load();
    return -1;
}

int swc_last_phase_ms(float* ms, int cap) { return ms ? last_phase_ms(ms, cap) : 0; }

// Everything the library holds for the next call goes back: the parked host results, the CALLING thread's page-locked staging
// buffers (they are thread-local: other threads release theirs by calling this themselves, or when they end), and -- after a
// synchronisation of the calling thread's stream -- the freed device memory in the current device's pool.
int swc_trim(void) try {
    {
        std::lock_guard<std::mutex> lk(g_res_mu);
        for (auto& e : g_res_parked) free(e.first);
        g_res_parked.clear();
    }
    for (int w = 0; w < 2; w++) (void)pinned_stage(w | 4, 0);
    int dev = 0;
    hipMemPool_t pool;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetDefaultMemPool(&pool, dev) == hipSuccess) {
        (void)hipStreamSynchronize(hipStreamPerThread);
        (void)hipMemPoolTrimTo(pool, 0);
    }
    (void)hipGetLastError();
    return SWC_OK;
} catch (...) {
    return SWC_E_DEVICE;
}

int swc_device_available(void) { return device_ready() ? 1 : 0; }
const char* swc_version(void) { return "swc-hip 0.1 (gfx950)"; }
void swc_free(void* p) { host_result_free(p); }

size_t swc_batch_workspace_bytes(int codec, size_t n_jobs, uint64_t max_out_cap) {
    (void)max_out_cap;
    switch (codec) {
        case SWC_CODEC_LZMA:
        case SWC_CODEC_LZMA2: return n_jobs * lzma_spill_bytes_per_job();  // the literal coders of every stream (LDS caches one of them) + the long-length trees; optional, see the header
        case SWC_CODEC_BZIP2_BLOCK: return n_jobs * bzip2_ws_bytes_per_job((size_t)max_out_cap + 64);  // L is never longer than the output
        case SWC_CODEC_DEFLATE: return n_jobs * inflate_ws_bytes_per_job(max_out_cap);  // match records + literal stream of phase 1 (lz_resolve.h)
        case SWC_CODEC_LZ4_BLOCK: return n_jobs * lz4_ws_bytes_per_job(max_out_cap);     // same layout (lz4_wave.h); without it blocks decode one per lane
        default: return 0;
    }
}

int swc_batch_decompress_ws(int codec, swc_job* jobs, size_t n, void* workspace, size_t workspace_bytes,
                            const swc_batch_opts* opts) try {
    if (!device_ready()) return SWC_E_DEVICE;
    if (n && !jobs) return SWC_E_INVALID_ARGUMENT;
    if (opts && opts->device >= 0 && hipSetDevice(opts->device) != hipSuccess) return SWC_E_DEVICE;
    hipStream_t stream = opts ? static_cast<hipStream_t>(opts->stream) : nullptr;
    hipError_t e = launch_codec(codec, reinterpret_cast<Job*>(jobs), n, workspace, workspace_bytes, stream);
    if (e == hipErrorInvalidValue) return SWC_E_INVALID_ARGUMENT;
    if (e != hipSuccess) return SWC_E_DEVICE;
    if (opts && opts->synchronize && hipStreamSynchronize(stream) != hipSuccess) return SWC_E_DEVICE;
    return SWC_OK;
} catch (...) {   // std::bad_alloc / length_error from a size taken from the input: never through the C boundary
    return SWC_E_DEVICE;
}

int swc_batch_crc32(const swc_job* jobs, size_t n, uint32_t* crcs, const swc_batch_opts* opts) try {
    if (!device_ready()) return SWC_E_DEVICE;
    if (n && (!jobs || !crcs)) return SWC_E_INVALID_ARGUMENT;
    if (opts && opts->device >= 0 && hipSetDevice(opts->device) != hipSuccess) return SWC_E_DEVICE;
    hipStream_t stream = opts ? static_cast<hipStream_t>(opts->stream) : nullptr;
    if (launch_crc32(reinterpret_cast<const Job*>(jobs), n, crcs, stream) != hipSuccess) return SWC_E_DEVICE;
    if (opts && opts->synchronize && hipStreamSynchronize(stream) != hipSuccess) return SWC_E_DEVICE;
    return SWC_OK;
} catch (...) {   // std::bad_alloc / length_error from a size taken from the input: never through the C boundary
    return SWC_E_DEVICE;
}

int swc_batch_checksum(int kind, const swc_job* jobs, size_t n, uint64_t* sums, const swc_batch_opts* opts) try {
    if (!device_ready()) return SWC_E_DEVICE;
    if (kind < SWC_SUM_CRC32 || kind > SWC_SUM_XXH32 || (n && (!jobs || !sums))) return SWC_E_INVALID_ARGUMENT;
    if (opts && opts->device >= 0 && hipSetDevice(opts->device) != hipSuccess) return SWC_E_DEVICE;
    hipStream_t stream = opts ? static_cast<hipStream_t>(opts->stream) : nullptr;
    if (launch_checksum(kind, reinterpret_cast<const Job*>(jobs), n, sums, stream) != hipSuccess) return SWC_E_DEVICE;
    if (opts && opts->synchronize && hipStreamSynchronize(stream) != hipSuccess) return SWC_E_DEVICE;
    return SWC_OK;
} catch (...) {   // std::bad_alloc / length_error from a size taken from the input: never through the C boundary
    return SWC_E_DEVICE;
}

int swc_batch_decompress(int codec, swc_job* jobs, size_t n, const swc_batch_opts* opts) try {
    if (codec == SWC_CODEC_BZIP2_BLOCK) return SWC_E_INVALID_ARGUMENT;  // needs the workspace: use swc_batch_decompress_ws
    if ((codec == SWC_CODEC_DEFLATE || codec == SWC_CODEC_LZ4_BLOCK) && n) {
        // Deflate and LZ4 need the record / literal workspace: every job gets an area sized from ITS capacity (prefix sums,
        // computed here from the job records) and the stream-ordered allocator owns it for the duration of the launch.
        if (!device_ready()) return SWC_E_DEVICE;
        if (!jobs) return SWC_E_INVALID_ARGUMENT;
        if (opts && opts->device >= 0 && hipSetDevice(opts->device) != hipSuccess) return SWC_E_DEVICE;
        hipStream_t stream = opts ? static_cast<hipStream_t>(opts->stream) : nullptr;
        std::vector<swc_job> host(n);
        if (hipMemcpyAsync(host.data(), jobs, n * sizeof(swc_job), hipMemcpyDeviceToHost, stream) != hipSuccess) return SWC_E_DEVICE;
        if (hipStreamSynchronize(stream) != hipSuccess) return SWC_E_DEVICE;
        std::vector<uint64_t> off(n + 1);
        uint64_t total = 0;
        for (size_t i = 0; i < n; i++) { off[i] = total; total += inflate_ws_bytes_per_job(host[i].out_cap); }
        off[n] = total;
        void* ws = nullptr;
        const size_t off_bytes = (n + 1) * sizeof(uint64_t);
        if (hipMallocAsync(&ws, total + off_bytes + 16, stream) != hipSuccess) { (void)hipGetLastError(); return SWC_E_DEVICE; }
        uint8_t* d_off = static_cast<uint8_t*>(ws) + ((total + 15) & ~(uint64_t)15);
        int st = SWC_OK;
        if (hipMemcpyAsync(d_off, off.data(), off_bytes, hipMemcpyHostToDevice, stream) != hipSuccess) st = SWC_E_DEVICE;
        if (st == SWC_OK && hipStreamSynchronize(stream) != hipSuccess) st = SWC_E_DEVICE;   // (`off` lives on this stack frame)
        if (st == SWC_OK && launch_codec(codec, reinterpret_cast<Job*>(jobs), n, ws, total, stream, reinterpret_cast<const uint64_t*>(d_off)) != hipSuccess) st = SWC_E_DEVICE;
        (void)hipFreeAsync(ws, stream);
        if (st == SWC_OK && opts && opts->synchronize && hipStreamSynchronize(stream) != hipSuccess) st = SWC_E_DEVICE;
        return st;
    }
    return swc_batch_decompress_ws(codec, jobs, n, nullptr, 0, opts);
} catch (...) {   // std::bad_alloc / length_error from a size taken from the input: never through the C boundary
    return SWC_E_DEVICE;
}

}  // extern "C"
