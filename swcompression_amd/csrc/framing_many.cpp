// framing_many.cpp -- container callers that batch through the engine.
//
//   swc_unarchive_many        many independent archives of one kind: block discovery for ALL of them on the host, then
//                             ONE batched launch, then the per-archive trailer checks in the reference's order.  The
//                             reference handles such a set one archive (and, inside it, one block) after the other.
//   swc_zip_get_entries_data  ZipContainer.getEntryData (reference Sources/ZIP/ZipContainer.swift:61-118) for every entry
//                             of a container at once: each entry is an independent Deflate / BZip2 / LZMA stream whose
//                             location and sizes the central directory already gave to the caller.
#include <algorithm>
#include <memory>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <deque>
#include <functional>
#include <chrono>
#include <cstdlib>
#include <map>
#include <vector>
#include "framing.h"

namespace swc {
namespace {

inline uint32_t le32(const uint8_t* p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24; }
inline uint64_t le64(const uint8_t* p) { return (uint64_t)le32(p) | (uint64_t)le32(p + 4) << 32; }

struct Result {
    int status = SWC_OK;
    std::vector<uint8_t> data;
};

void hand_over(const std::vector<Result>& res, uint8_t** outs, size_t* out_lens, int32_t* statuses) {
    for (size_t i = 0; i < res.size(); i++) {
        give(res[i].data, &outs[i], &out_lens[i]);
        statuses[i] = res[i].status;
    }
}

// kinds 1-3 and 7: one unit per archive
int many_single_unit(int kind, const uint8_t* const* archives, const size_t* lens, size_t n, std::vector<Result>& res) {
    const int codec = kind == 7 ? SWC_CODEC_LZMA2 : SWC_CODEC_DEFLATE;
    std::vector<HostUnit> units;
    std::vector<size_t> owner, data_pos;
    for (size_t i = 0; i < n; i++) {
        const uint8_t* d = archives[i];
        const size_t len = lens[i];
        HostUnit u;
        size_t p = 0;
        int st = SWC_OK;
        switch (kind) {
            case 1: st = gzip_member_prepare(d, len, 0, u); p = st ? 0 : (size_t)(u.in - d); break;   // GzipArchive.swift:38-48
            case 2: st = zlib_parse_header(d, len, p); u.in = d + p; u.in_len = len - p; break;       // ZlibArchive.swift:25-31
            case 3: u.in = d; u.in_len = len; break;                                                  // Deflate.swift:24-28
            default:                                                                                  // LZMA2.swift:25-30
                if (len < 1) { st = SWC_E_LZMA_RANGE_DECODER_INIT_ERROR; break; }
                u.in = d + 1; u.in_len = len - 1; u.aux = d[0];
                u.cap_hint = std::max<size_t>(lzma2_announced_size(d + 1, len - 1), 16);
        }
        if (st) { res[i].status = st; continue; }
        units.push_back(std::move(u));
        owner.push_back(i);
        data_pos.push_back(p);
    }
    if (!units.empty()) {
        int st = run_units(codec, units);
        if (st) return st;
    }
    for (size_t k = 0; k < units.size(); k++) {
        HostUnit& u = units[k];
        Result& r = res[owner[k]];
        const uint8_t* d = archives[owner[k]];
        const size_t len = lens[owner[k]];
        if (kind == 1) {
            size_t next;
            bool crc_error;
            r.status = gzip_member_finish(d, len, data_pos[k], u, next, crc_error);
            if (r.status) continue;
            r.data = std::move(u.out);                                       // wrongCRC carries the member (:44)
            if (crc_error) r.status = SWC_E_GZIP_WRONG_CRC;
        } else if (kind == 2) {
            if (u.status) { r.status = u.status; continue; }
            const size_t q = data_pos[k] + u.in_consumed;
            r.data = std::move(u.out);                                       // wrongAdler32 carries the data (:34,39)
            if (len - q < 4) { r.status = SWC_E_ZLIB_WRONG_ADLER32; continue; }
            const uint32_t stored = (uint32_t)d[q] << 24 | (uint32_t)d[q + 1] << 16 | (uint32_t)d[q + 2] << 8 | d[q + 3];
            r.status = swc_adler32(r.data.data(), r.data.size()) == stored ? SWC_OK : SWC_E_ZLIB_WRONG_ADLER32;
        } else {
            r.status = u.status;
            if (!u.status) r.data = std::move(u.out);
        }
    }
    return SWC_OK;
}

// kind 4: LZ4.decompress(data:) per archive; archives that open with a frame of independent blocks share one launch
int many_lz4(const uint8_t* const* archives, const size_t* lens, size_t n, std::vector<Result>& res) {
    std::vector<HostUnit> units;
    std::vector<std::unique_ptr<Lz4Plan>> plans(n);
    for (size_t i = 0; i < n; i++) {
        plans[i].reset(new Lz4Plan);
        if (!lz4_plan_prepare(archives[i], lens[i], *plans[i], units)) plans[i].reset();
    }
    if (!units.empty()) {
        int st = run_units(SWC_CODEC_LZ4_BLOCK, units);
        if (st) return st;
    }
    for (size_t i = 0; i < n; i++) {
        if (plans[i]) {
            res[i].status = lz4_plan_finish(archives[i], lens[i], *plans[i], units, res[i].data);
        } else {   // skippable / legacy / dependent-block frames: the single-archive path
            uint8_t* o = nullptr;
            size_t ol = 0, used = 0;
            res[i].status = swc_lz4_decompress(archives[i], lens[i], nullptr, 0, -1, &o, &ol, &used);
            if (res[i].status == SWC_E_DEVICE) { swc_free(o); return SWC_E_DEVICE; }
            res[i].data.assign(o, o + ol);
            swc_free(o);
        }
    }
    return SWC_OK;
}

// kind 5: BZip2.decompress(data:) per archive; the candidate blocks of ALL streams share one launch
int many_bzip2(const uint8_t* const* archives, const size_t* lens, size_t n, std::vector<Result>& res) {
    std::vector<HostUnit> units;
    std::vector<std::vector<uint64_t>> used(n);
    std::vector<size_t> first(n);
    for (size_t i = 0; i < n; i++) {
        first[i] = units.size();
        bzip2_collect_candidates(archives[i], lens[i], units, used[i]);
    }
    if (!units.empty()) {
        int st = run_units(SWC_CODEC_BZIP2_BLOCK, units);
        if (st) return st;
    }
    for (size_t i = 0; i < n; i++) {
        size_t pos = 0;
        res[i].status = bzip2_finish_stream(archives[i], lens[i], units, first[i], used[i], res[i].data, pos);
        if (res[i].status == SWC_E_DEVICE) return SWC_E_DEVICE;
        if (res[i].status != SWC_OK && res[i].status != SWC_E_BZIP2_WRONG_CRC) res[i].data.clear();   // only wrongCRC carries data
    }
    return SWC_OK;
}

// kind 6: XZArchive.unarchive per archive.  Block boundaries of an .xz stream are only known block by block (or from the
// index at its end), so every archive takes the single-archive path: one launch per block.
int many_xz(const uint8_t* const* archives, const size_t* lens, size_t n, std::vector<Result>& res) {
    for (size_t i = 0; i < n; i++) {
        uint8_t* o = nullptr;
        size_t ol = 0;
        res[i].status = swc_xz_unarchive(archives[i], lens[i], &o, &ol);
        if (res[i].status == SWC_E_DEVICE) { swc_free(o); return SWC_E_DEVICE; }
        res[i].data.assign(o, o + ol);
        swc_free(o);
    }
    return SWC_OK;
}

int many_dispatch(int kind, const uint8_t* const* archives, const size_t* lens, size_t n, std::vector<Result>& res) {
    switch (kind) {
        case 4: return many_lz4(archives, lens, n, res);
        case 5: return many_bzip2(archives, lens, n, res);
        case 6: return many_xz(archives, lens, n, res);
        default: return many_single_unit(kind, archives, lens, n, res);
    }
}

}  // namespace
}  // namespace swc

using namespace swc;

// swc_unarchive_many over several GPUs of one node: the archives are independent, so the list is cut into one contiguous
// range per device, balanced by compressed + declared uncompressed bytes (sum(C + U); U where the framing
// declares it -- the ISIZE of a gzip member, the content size of an LZ4 frame -- else C stands for both), and every range runs
// the single-device path on that device's WORKER thread with the device current (stream, staging buffers and workspace are
// per thread, api.cpp).  The workers live as long as the process, one per device ever named: a call does not pay for thread
// start-up, and the page-locked staging buffers of a device are allocated once, not per call (round-2 review).  No
// data-path exchange between devices: results are simply placed at their archive's index.
namespace {
struct DeviceWorker {
    std::mutex m;
    std::condition_variable cv, idle_cv;
    std::deque<std::function<void()>> q;
    bool busy = false;
    int device;
    explicit DeviceWorker(int dev) : device(dev) {
        std::thread([this] {
            for (;;) {
                std::function<void()> f;
                {
                    std::unique_lock<std::mutex> lk(m);
                    cv.wait(lk, [this] { return !q.empty(); });
                    f = std::move(q.front());
                    q.pop_front();
                    busy = true;
                }
                f();
                f = nullptr;   // (the task's captures go before the worker reports itself idle)
                { std::lock_guard<std::mutex> lk(m); busy = false; if (q.empty()) idle_cv.notify_all(); }
            }
        }).detach();   // (ends with the process: a joinable static thread would have to outlive the HIP runtime's own teardown)
    }
    void post(std::function<void()> f) {
        { std::lock_guard<std::mutex> lk(m); q.push_back(std::move(f)); }
        cv.notify_one();
    }
    // waits (bounded) until nothing is queued or running
    void drain() {
        std::unique_lock<std::mutex> lk(m);
        idle_cv.wait_for(lk, std::chrono::seconds(10), [this] { return q.empty() && !busy; });
    }
};
struct WorkerPool {
    std::mutex m;
    std::map<int, DeviceWorker*> pool;   // (never destroyed: the threads are detached)
    static WorkerPool& get() {
        static WorkerPool* p = [] {
            WorkerPool* w = new WorkerPool;
            // exit() while another thread is still inside a call: let its tasks finish before the HIP runtime and the statics go
            std::atexit([] { WorkerPool& wp = WorkerPool::get(); std::vector<DeviceWorker*> ws; { std::lock_guard<std::mutex> lk(wp.m); for (auto& kv : wp.pool) ws.push_back(kv.second); } for (DeviceWorker* w2 : ws) w2->drain(); });
            return w;
        }();
        return *p;
    }
};
DeviceWorker& worker_of(int device) {
    WorkerPool& wp = WorkerPool::get();
    std::lock_guard<std::mutex> lk(wp.m);
    auto it = wp.pool.find(device);
    if (it == wp.pool.end()) it = wp.pool.emplace(device, new DeviceWorker(device)).first;
    return *it->second;
}
// the uncompressed size the framing declares, 0 if it declares none (or nonsense: a hint for the balance only)
uint64_t declared_size(int kind, const uint8_t* p, size_t len) {
    if (kind == 1 && len >= 18) return (uint64_t)p[len - 4] | ((uint64_t)p[len - 3] << 8) | ((uint64_t)p[len - 2] << 16) | ((uint64_t)p[len - 1] << 24);   // ISIZE (GzipArchive.swift:94)
    if (kind == 4 && len >= 15 && p[0] == 0x04 && p[1] == 0x22 && p[2] == 0x4D && p[3] == 0x18 && (p[4] & 0x08)) {   // LZ4 frame with the content-size field (LZ4.swift:205-213)
        uint64_t v = 0;
        for (int i = 7; i >= 0; i--) v = (v << 8) | p[6 + i];
        return v < ((uint64_t)1 << 40) ? v : 0;
    }
    return 0;
}
}  // namespace


extern "C" {

int swc_unarchive_many(int kind, const uint8_t* const* archives, const size_t* lens, size_t n,
                       uint8_t** outs, size_t* out_lens, int32_t* statuses) try {
    if (kind < 1 || kind > 7 || (n && (!archives || !lens || !outs || !out_lens || !statuses))) return SWC_E_INVALID_ARGUMENT;
    for (size_t i = 0; i < n; i++) if (lens[i] && !archives[i]) return SWC_E_INVALID_ARGUMENT;
    if (!device_ready()) return SWC_E_DEVICE;
    std::vector<Result> res(n);
    const int st = many_dispatch(kind, archives, lens, n, res);
    if (st) return st;
    hand_over(res, outs, out_lens, statuses);
    return SWC_OK;
} catch (...) {   // std::bad_alloc / length_error from a size taken from the input: never through the C boundary
    for (size_t i = 0; i < n; i++) { outs[i] = host_result(0); out_lens[i] = 0; statuses[i] = SWC_E_DEVICE; }   // (results are handed over last: nothing of theirs is lost)
    return SWC_E_DEVICE;
}

int swc_unarchive_many_devices(int kind, const uint8_t* const* archives, const size_t* lens, size_t n,
                               const int* devices, size_t n_devices, uint8_t** outs, size_t* out_lens, int32_t* statuses) try {
    if (kind < 1 || kind > 7 || (n && (!archives || !lens || !outs || !out_lens || !statuses)) || !devices || n_devices == 0)
        return SWC_E_INVALID_ARGUMENT;
    for (size_t i = 0; i < n; i++) if (lens[i] && !archives[i]) return SWC_E_INVALID_ARGUMENT;
    // only the LISTED devices matter (the calling thread's current device may be something else altogether)
    for (size_t d = 0; d < n_devices; d++) if (!device_is_gfx950(devices[d])) return SWC_E_DEVICE;
    // contiguous ranges of (about) equal cost C + U; +1 per archive so that empty inputs spread as well
    std::vector<size_t> cut(n_devices + 1, n);
    cut[0] = 0;
    {
        std::vector<uint64_t> cost(n);
        unsigned __int128 total = 0;
        for (size_t i = 0; i < n; i++) {
            const uint64_t u = declared_size(kind, archives[i], lens[i]);
            cost[i] = (uint64_t)lens[i] + (u ? u : (uint64_t)lens[i]) + 1;
            total += cost[i];
        }
        unsigned __int128 acc = 0;
        size_t d = 1;
        for (size_t i = 0; i < n && d < n_devices; i++) {
            acc += cost[i];
            while (d < n_devices && acc * n_devices >= total * d) cut[d++] = i + 1;
        }
    }
    // Everything a task touches lives in ONE shared block that the tasks hold by value: the caller waits for all of them, but
    // a worker may still be inside notify / unlock when the caller wakes up, and must not find the frame gone.
    struct Shared {
        std::vector<Result> res;
        std::vector<int> st;
        std::mutex m;
        std::condition_variable cv;
        size_t outstanding = 0;
    };
    auto sh = std::make_shared<Shared>();
    sh->res.resize(n);
    sh->st.assign(n_devices, SWC_OK);
    bool post_failed = false;
    for (size_t d = 0; d < n_devices && !post_failed; d++) {
        const size_t lo = cut[d], hi = cut[d + 1];
        if (lo >= hi) continue;
        const int dev = devices[d];
        try {
            { std::lock_guard<std::mutex> lk(sh->m); sh->outstanding++; }
            worker_of(dev).post([sh, kind, archives, lens, d, lo, hi, dev] {
                try {
                    // (per task, not once per thread: a failed or later reset device must not let the range run elsewhere)
                    if (hipSetDevice(dev) != hipSuccess) sh->st[d] = SWC_E_DEVICE;
                    else {
                        std::vector<Result> part(hi - lo);
                        sh->st[d] = many_dispatch(kind, archives + lo, lens + lo, hi - lo, part);
                        if (sh->st[d] == SWC_OK)
                            for (size_t i = lo; i < hi; i++) sh->res[i] = std::move(part[i - lo]);
                    }
                } catch (...) {
                    sh->st[d] = SWC_E_DEVICE;
                }
                std::lock_guard<std::mutex> lk(sh->m);   // (notify under the lock: the waiter cannot leave between the two)
                sh->outstanding--;
                sh->cv.notify_all();
            });
        } catch (...) {   // worker_of / post could not allocate: this range was never queued
            { std::lock_guard<std::mutex> lk(sh->m); sh->outstanding--; }
            sh->st[d] = SWC_E_DEVICE;
            post_failed = true;
        }
    }
    {   // every task that WAS posted is waited for, whatever happened above: they read the caller's archives
        std::unique_lock<std::mutex> lk(sh->m);
        sh->cv.wait(lk, [&] { return sh->outstanding == 0; });
    }
    std::vector<Result>& res = sh->res;
    std::vector<int>& st = sh->st;
    for (size_t d = 0; d < n_devices; d++) if (st[d]) return st[d];
    hand_over(res, outs, out_lens, statuses);
    return SWC_OK;
} catch (...) {
    for (size_t i = 0; i < n; i++) { outs[i] = host_result(0); out_lens[i] = 0; statuses[i] = SWC_E_DEVICE; }
    return SWC_E_DEVICE;
}

int swc_zip_get_entries_data(const uint8_t* container, size_t len, swc_zip_entry* entries, size_t n) try {
    if ((len && !container) || (n && !entries)) return SWC_E_INVALID_ARGUMENT;
    if (!device_ready()) return SWC_E_DEVICE;
    // ---- units per compression method (ZipContainer.swift:70-93)
    std::vector<HostUnit> defl, lzma;
    std::vector<size_t> defl_owner, lzma_owner;
    std::vector<Result> res(n);
    std::vector<size_t> real_comp(n, 0);
    for (size_t i = 0; i < n; i++) { entries[i].data = nullptr; entries[i].data_len = 0; }
    for (size_t i = 0; i < n; i++) {
        swc_zip_entry& e = entries[i];
        e.data = nullptr; e.data_len = 0; e.crc_error = 0; e.status = SWC_OK;
        if (e.data_offset > len) { res[i].status = SWC_E_REF_TRAP; continue; }          // reader offset past the end: trap
        const uint8_t* p = container + e.data_offset;
        const size_t avail = len - (size_t)e.data_offset;
        switch (e.method) {
            case 0:                                                                     // .copy :70-71
                if (e.uncomp_size > avail) { res[i].status = SWC_E_REF_TRAP; break; }   // bytes(count:) past the end
                res[i].data.assign(p, p + e.uncomp_size);
                real_comp[i] = (size_t)e.uncomp_size;
                break;
            case 8: {                                                                   // .deflate :72-78
                HostUnit u;
                u.in = p; u.in_len = avail;                                             // the reader runs on into whatever follows (:74)
                u.base = container; u.base_len = len;                                   // ... but the container is staged once for all entries
                if (e.uncomp_size <= (uint64_t)avail * 1100 + 4096) u.cap_hint = std::max<size_t>((size_t)e.uncomp_size, 64);
                defl.push_back(std::move(u));
                defl_owner.push_back(i);
                break;
            }
            case 12: {                                                                  // .bzip2 :79-86 (one stream: its own discovery)
                uint8_t* o = nullptr;
                size_t ol = 0, used = 0;
                // first over the entry's own bytes (the block-magic scan and the decode of every candidate behind them
                // would otherwise cover the rest of the container for every entry); anything but a clean decode that ends
                // inside them is decided by a second run over the whole tail, as the reference's reader would see it
                const bool sized = !e.has_data_descriptor && e.comp_size > 0 && e.comp_size < avail;
                int st = swc_bzip2_decompress(p, sized ? (size_t)e.comp_size : avail, &o, &ol, &used);
                if (sized && st != SWC_OK && st != SWC_E_DEVICE) {
                    swc_free(o);
                    o = nullptr;
                    st = swc_bzip2_decompress(p, avail, &o, &ol, &used);
                }
                if (st == SWC_E_DEVICE) { swc_free(o); return SWC_E_DEVICE; }
                if (st == SWC_OK) res[i].data.assign(o, o + ol);
                swc_free(o);
                res[i].status = st;                                                      // BZip2Error propagates (wrongCRC too: try, no catch)
                real_comp[i] = used;
                break;
            }
            case 14: {                                                                  // .lzma :87-89
                if (avail < 9) { res[i].status = SWC_E_REF_TRAP; break; }               // 4 skipped bytes + properties: reads past the end trap
                const uint32_t b = p[4];
                if (b >= 225) { res[i].status = SWC_E_LZMA_WRONG_PROPERTIES; break; }   // LZMAProperties.swift:51
                if (e.uncomp_size > (uint64_t)INT64_MAX) { res[i].status = SWC_E_REF_TRAP; break; }  // toInt() traps
                HostUnit u;
                u.in = p + 9; u.in_len = avail - 9;
                u.base = container; u.base_len = len;
                u.aux = (int32_t)((b % 9) | (((b / 9) % 5) << 8) | (((b / 9) / 5) << 16));
                u.extra = e.uncomp_size;
                u.dict_value = le32(p + 5);
                u.cap_hint = (size_t)e.uncomp_size + 16;
                lzma.push_back(std::move(u));
                lzma_owner.push_back(i);
                break;
            }
            default: res[i].status = SWC_E_ZIP_COMPRESSION_NOT_SUPPORTED;               // :90-91
        }
    }
    if (!defl.empty() && run_units(SWC_CODEC_DEFLATE, defl) != SWC_OK) return SWC_E_DEVICE;
    if (!lzma.empty() && run_units(SWC_CODEC_LZMA, lzma) != SWC_OK) return SWC_E_DEVICE;
    for (size_t k = 0; k < defl.size(); k++) {
        const size_t i = defl_owner[k];
        res[i].status = defl[k].status;
        if (!defl[k].status) { res[i].data = std::move(defl[k].out); real_comp[i] = defl[k].in_consumed; }
    }
    for (size_t k = 0; k < lzma.size(); k++) {
        const size_t i = lzma_owner[k];
        res[i].status = lzma[k].status;
        if (!lzma[k].status) { res[i].data = std::move(lzma[k].out); real_comp[i] = 9 + lzma[k].in_consumed; }
    }
    // ---- data descriptor, size and CRC checks (:94-117)
    for (size_t i = 0; i < n; i++) {
        swc_zip_entry& e = entries[i];
        Result& r = res[i];
        if (r.status == SWC_OK) {
            uint64_t comp = e.comp_size, uncomp = e.uncomp_size;
            uint32_t crc = e.crc32;
            if (e.has_data_descriptor) {
                size_t q = (size_t)e.data_offset + real_comp[i];
                const size_t need = 4 + 4 + (e.zip64 ? 16 : 8);
                if (len - q < 4) r.status = SWC_E_REF_TRAP;
                else {
                    if (le32(container + q) == 0x08074b50u) q += 4;                     // optional signature :98-101
                    if (len < q || len - q < need - 4) r.status = SWC_E_REF_TRAP;
                    else {
                        crc = le32(container + q);
                        if (e.zip64) { comp = le64(container + q + 4); uncomp = le64(container + q + 12); }
                        else { comp = le32(container + q + 4); uncomp = le32(container + q + 8); }
                    }
                }
            }
            if (r.status == SWC_OK && !(comp == (uint64_t)real_comp[i] && uncomp == (uint64_t)r.data.size())) r.status = SWC_E_ZIP_WRONG_SIZE;  // :112-113
            if (r.status == SWC_OK) e.crc_error = crc != swc_crc32(r.data.data(), r.data.size(), 0) ? 1 : 0;  // :114
        }
        e.status = r.status;
        if (r.status != SWC_OK) r.data.clear();
        size_t dl = 0;
        give(r.data, &e.data, &dl);
        e.data_len = dl;
    }
    return SWC_OK;
} catch (...) {   // std::bad_alloc / length_error from a size taken from the input: never through the C boundary
    for (size_t i = 0; i < n; i++) { entries[i].status = SWC_E_DEVICE; if (!entries[i].data) entries[i].data_len = 0; }
    return SWC_E_DEVICE;
}

// Host block discovery as a library call: what the batched entry points use internally, for callers
// that keep their data on the device and build their own job lists.  No device needed.
int swc_index_blocks(int kind, const uint8_t* in, size_t len, swc_block_ref* refs, size_t cap, size_t* n) try {
    if (!n || (len && !in) || (cap && !refs)) return SWC_E_INVALID_ARGUMENT;
    std::vector<BlockRef64> v;
    bool ok = true;
    int st = SWC_OK;
    switch (kind) {
        case 1: ok = bgzf_index(in, len, v); break;
        case 4: ok = lz4_frame_index(in, len, v); break;
        case 5: bzip2_magic_index(in, len, v); break;
        case 6: xz_block_index(in, len, v); break;
        case 7: st = lzma2_chunk_index(in, len, v); break;
        default: return SWC_E_INVALID_ARGUMENT;
    }
    if (!ok) { *n = 0; return SWC_E_INVALID_ARGUMENT; }
    *n = v.size();
    for (size_t i = 0; i < v.size() && i < cap; i++) refs[i] = swc_block_ref{v[i].offset, v[i].comp_len, v[i].uncomp_len, v[i].aux, v[i].flags};
    return st;
} catch (...) {   // std::bad_alloc / length_error from a size taken from the input: never through the C boundary
    if (n) *n = 0;
    return SWC_E_DEVICE;
}

// SevenZipFolder.unpack(data:) for many folders (reference Sources/7-Zip/7zFolder.swift:138-194).  Stage k of every
// chain that is still alive is decoded together: one run_units launch per codec and stage.
int swc_7z_unpack_folders(swc_7z_folder* folders, size_t n) try {
    if (n && !folders) return SWC_E_INVALID_ARGUMENT;
    size_t max_chain = 0;
    for (size_t i = 0; i < n; i++) {
        if ((folders[i].len && !folders[i].data) || (folders[i].n_coders && !folders[i].coders)) return SWC_E_INVALID_ARGUMENT;
        max_chain = std::max(max_chain, folders[i].n_coders);
        folders[i].status = SWC_OK; folders[i].out = nullptr; folders[i].out_len = 0;
    }
    if (!device_ready()) return SWC_E_DEVICE;
    std::vector<Result> cur(n);          // decodedData of :139; stage 0 reads the packed stream itself
    std::vector<char> own(n, 0);         // cur[i].data holds the data (else: still the caller's buffer)
    auto in_ptr = [&](size_t i) { return own[i] ? cur[i].data.data() : folders[i].data; };
    auto in_len = [&](size_t i) { return own[i] ? cur[i].data.size() : folders[i].len; };
    for (size_t stage = 0; stage < max_chain; stage++) {
        std::vector<size_t> by_method[7];
        for (size_t i = 0; i < n; i++) {
            if (cur[i].status != SWC_OK || stage >= folders[i].n_coders) continue;
            const swc_7z_coder& c = folders[i].coders[stage];
            if (c.multi_stream) { cur[i].status = SWC_E_7Z_MULTI_STREAM_NOT_SUPPORTED; continue; }          // :141-142
            switch (c.method) {
                case 0: break;                                                                             // .copy: continue (:147-148), no size check
                case 1: case 2: case 6: by_method[c.method].push_back(i); break;
                case 3: if (c.props_len != 1) cur[i].status = SWC_E_LZMA2_WRONG_DICTIONARY_SIZE;            // :155-157
                        else by_method[3].push_back(i);
                        break;
                case 4: if (c.props_len != 5) cur[i].status = SWC_E_LZMA_WRONG_PROPERTIES;                  // :162-164
                        else if (c.props[0] >= 225) cur[i].status = SWC_E_LZMA_WRONG_PROPERTIES;            // LZMAProperties.swift:51
                        else by_method[4].push_back(i);
                        break;
                case 5: if (c.props_len != 1) cur[i].status = SWC_E_7Z_INTERNAL_STRUCTURE_ERROR;            // :177-179
                        else by_method[5].push_back(i);
                        break;
                case 7: cur[i].status = SWC_E_7Z_ENCRYPTION_NOT_SUPPORTED; break;                           // :185
                default: cur[i].status = SWC_E_7Z_COMPRESSION_NOT_SUPPORTED;                                // :187
            }
        }
        std::vector<Result> next(n);
        std::vector<char> done(n, 0);
        // Deflate / LZMA2 / LZMA: one unit per folder
        for (int m : {1, 3, 4}) {
            const std::vector<size_t>& idx = by_method[m];
            if (idx.empty()) continue;
            std::vector<HostUnit> units(idx.size());
            for (size_t k = 0; k < idx.size(); k++) {
                const size_t i = idx[k];
                const swc_7z_coder& c = folders[i].coders[stage];
                HostUnit& u = units[k];
                u.in = in_ptr(i); u.in_len = in_len(i);
                u.cap_hint = std::max<size_t>((size_t)c.unpack_size + 16, 64);
                if (m == 3) u.aux = c.props[0];
                if (m == 4) {
                    const uint32_t b = c.props[0];
                    u.aux = (int32_t)((b % 9) | (((b / 9) % 5) << 8) | (((b / 9) / 5) << 16));
                    u.extra = c.unpack_size;                                                               // uncompressedSize: unpackSize :172
                    u.dict_value = (uint64_t)c.props[1] | (uint64_t)c.props[2] << 8 | (uint64_t)c.props[3] << 16 | (uint64_t)c.props[4] << 24;
                }
            }
            if (run_units(m == 1 ? SWC_CODEC_DEFLATE : m == 3 ? SWC_CODEC_LZMA2 : SWC_CODEC_LZMA, units) != SWC_OK) return SWC_E_DEVICE;
            for (size_t k = 0; k < idx.size(); k++) {
                next[idx[k]].status = units[k].status;
                if (!units[k].status) next[idx[k]].data = std::move(units[k].out);
                done[idx[k]] = 1;
            }
        }
        // BZip2 / LZ4: block discovery per folder, shared launch (the many-archive paths above)
        for (int m : {2, 6}) {
            const std::vector<size_t>& idx = by_method[m];
            if (idx.empty()) continue;
            std::vector<const uint8_t*> ptrs(idx.size());
            std::vector<size_t> lens(idx.size());
            for (size_t k = 0; k < idx.size(); k++) { ptrs[k] = in_ptr(idx[k]); lens[k] = in_len(idx[k]); }
            std::vector<Result> r(idx.size());
            const int st = m == 2 ? many_bzip2(ptrs.data(), lens.data(), idx.size(), r) : many_lz4(ptrs.data(), lens.data(), idx.size(), r);
            if (st) return st;
            for (size_t k = 0; k < idx.size(); k++) {
                if (r[k].status) r[k].data.clear();          // errors thrown out of unpack() carry nothing here
                next[idx[k]] = std::move(r[k]);
                done[idx[k]] = 1;
            }
        }
        // Delta filter (DeltaFilter.swift:11-33) on the host
        for (size_t i : by_method[5]) {
            const int distance = (uint8_t)(folders[i].coders[stage].props[0] + 1);                          // properties[0] &+ 1 (:181)
            const uint8_t* src = in_ptr(i);
            const size_t len = in_len(i);
            uint8_t delta[256] = {0};
            int pos = 0;
            next[i].data.resize(len);
            for (size_t k = 0; k < len; k++) {
                const uint8_t tmp = (uint8_t)(src[k] + delta[(distance + pos) % 256]);
                delta[pos] = tmp;
                next[i].data[k] = tmp;
                pos = pos == 0 ? 255 : pos - 1;
            }
            done[i] = 1;
        }
        for (size_t i = 0; i < n; i++) {
            if (!done[i]) continue;
            if (next[i].status == SWC_OK && next[i].data.size() != folders[i].coders[stage].unpack_size) next[i].status = SWC_E_7Z_WRONG_SIZE;  // :190-191
            cur[i] = std::move(next[i]);
            own[i] = 1;
        }
    }
    for (size_t i = 0; i < n; i++) {
        folders[i].status = cur[i].status;
        if (cur[i].status != SWC_OK) { give_empty(&folders[i].out, &folders[i].out_len); continue; }
        if (own[i]) give(cur[i].data, &folders[i].out, &folders[i].out_len);
        else { std::vector<uint8_t> copy(folders[i].data, folders[i].data + folders[i].len); give(copy, &folders[i].out, &folders[i].out_len); }
    }
    return SWC_OK;
} catch (...) {   // std::bad_alloc / length_error from a size taken from the input: never through the C boundary
    return SWC_E_DEVICE;
}

}  // extern "C"
