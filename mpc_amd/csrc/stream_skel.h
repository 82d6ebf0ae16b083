// stream_skel.h — the streaming evaluator's byte skeletons (a block seen before is recognised without decoding a gate) and
// the helper threads of the match (stream_eval.cpp).
#pragma once

#include <condition_variable>
#include <mutex>
#include <thread>

#include "skel_match.h"
#include "stream_internal.h"

namespace gcs {

// Byte skeleton of a block the evaluator has parsed before.  A stream repeats its circuits (every call of a function
// compiles to the same block up to the global wires it is bound to), and the garbler serialises a given circuit the same
// way every time: the same op / flag bytes and tmp ids at the same offsets, only the table rows and the global ids
// differ.  A new block that equals a skeleton on every other byte, and whose global ids repeat in the same pattern, IS
// that circuit: no gate is decoded, the rows are copied out, the global ids are read at their known offsets.
struct EvalSkel {
    struct Chunk {       // what the parser records on its way through a block
        uint32_t cmp;    // bytes that must equal the reference block
        uint16_t skip;   // then a global id field (2 / 4 bytes), 0: none
        uint16_t nrows;  // then table rows (16 bytes each)
    };
    size_t nbytes = 0;
    uint32_t nrows = 0, nin = 0, nout = 0;
    CircEntry *ent = nullptr;                // owned by the stream's circuit cache (dropped with it on eviction)
    std::vector<uint8_t> bytes;              // the reference block ...
    std::vector<uint8_t> mask;               // ... and which of its bytes a block of this circuit must repeat (0xff / 0)
    std::vector<uint32_t> row_off;           // byte offset of every table row, in stream order
    std::vector<Chunk> chunks;               // (only in the parser's recording skeleton: build() turns it into mask + row_off)
    std::vector<uint32_t> gf_off;            // global id fields in stream order: byte offset | 1 << 31 for 4-byte ids
    std::vector<uint32_t> gf_canon;          // index of the first field that names the same wire (the repeat pattern)
    std::vector<uint32_t> in_gf, out_gf;     // field of input k (its first read) / of the k-th global write
    std::vector<uint8_t> out_live;           // 0: a later gate of the block writes the same wire (streaming.Set: last wins)
    int dev_index = -1;                      // its copy in the device-side matcher's table (stream_eval_dev.cpp; -1: none)
    const uint32_t *d_row_off = nullptr;     // ... and its row offsets there (a device array)
    uint32_t n_uniq = 0;                     // distinct global wires the block names; which of them input / output k is (the
    std::vector<uint32_t> in_u, out_u;       // device's verdict brings the distinct ids in order of first occurrence)
    uint64_t key = 0;                        // gates << 32 | tmp wires: what the skeletons are looked up by
    // what a block must equal byte-wise, as a hash: the reference bytes under the mask, the mask, the id fields, the row offsets.
    // Skeletons of ONE shape differ only in how the global ids repeat (an adder bound to another constant): a byte-wise
    // verdict on one of them holds for all, and only the cheap check of the repeat pattern has to be done per candidate.
    uint64_t shape = 0;
    void shape_hash() {
        uint64_t h = 1469598103934665603ull ^ bytes.size();
        auto mix = [&](uint64_t v) { h = (h ^ v) * 1099511628211ull; };
        for (size_t i = 0; i < bytes.size(); i++) mix((uint64_t)(bytes[i] & mask[i]) | ((uint64_t)mask[i] << 8));
        for (uint32_t f : gf_off) mix(f);
        for (uint32_t r : row_off) mix(0x100000000ull | r);
        shape = h ^ (h >> 31);
    }
    // the block cut into segments of about equal length, so that several threads can compare / copy side by side (SkelPool)
    struct Seg {
        size_t b0, b1;     // bytes [b0, b1)
        uint32_t r0, r1;   // rows [r0, r1): the rows that start inside [b0, b1)
    };
    std::vector<Seg> segs;
    size_t held() const { return bytes.size() + mask.size() + row_off.size() * sizeof(uint32_t); }
    // mask, row offsets and segments from the parser's chunk list (bytes already holds the block)
    void build(const std::vector<Chunk> &ch) {
        mask.assign(bytes.size(), 0);
        row_off.clear();
        segs.clear();
        const uint32_t n = (uint32_t)ch.size(), per = n / 8 >= 2048 ? n / 8 : n ? n : 1;
        size_t off = 0;
        for (uint32_t c = 0; c < n; c++) {
            if (c % per == 0 && (segs.empty() || n - c >= per / 2)) {
                if (!segs.empty()) segs.back().b1 = off, segs.back().r1 = (uint32_t)row_off.size();
                segs.push_back(Seg{off, bytes.size(), (uint32_t)row_off.size(), 0});
            }
            std::memset(mask.data() + off, 0xff, ch[c].cmp);
            off += (size_t)ch[c].cmp + ch[c].skip;
            for (uint32_t r = 0; r < ch[c].nrows; r++, off += 16) row_off.push_back((uint32_t)off);
        }
        if (segs.empty()) segs.push_back(Seg{0, bytes.size(), 0, 0});
        segs.back().b1 = bytes.size(), segs.back().r1 = (uint32_t)row_off.size();
    }
    // segment sg of `buf` against the reference: equal outside the global ids and the rows?  rows -> slab
    bool match_seg(const Seg &sg, const uint8_t *buf, gc_label *slab) const {
        if (!skel_simd::same(buf + sg.b0, bytes.data() + sg.b0, mask.data() + sg.b0, sg.b1 - sg.b0)) return false;
        if (slab) skel_simd::rows(buf, row_off.data() + sg.r0, sg.r1 - sg.r0, slab + sg.r0);
        return true;
    }
    // the table rows of a block that matched this skeleton with slab == nullptr, into dst (host order)
    void copy_rows(const uint8_t *buf, gc_label *dst) const { skel_simd::rows(buf, row_off.data(), row_off.size(), dst); }
};

// Helper threads for the skeleton match of a big block (a 131 072-gate block is ~33 000 compare runs and 43 000 rows: 0.42 ms
// on one core, more than the GPU needs for the block).  Segments are handed out through a counter; the calling thread
// takes its share.  GC_STREAM_THREADS = helpers (default 3, 0: none).
struct SkelPool {
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    // the job on offer (written under mu; a worker copies it under mu when it wakes up)
    struct Job {
        const EvalSkel *sk = nullptr;
        const uint8_t *buf = nullptr;
        gc_label *slab = nullptr;
        uint32_t id = 0;
    } job;
    bool stop = false;
    // next segment to hand out, tagged with the job it belongs to (id << 32 | index): a worker that was descheduled between
    // its last segment of job k and its next look at the counter must not take — or skip — a segment of job k + 1
    std::atomic<uint64_t> next{0};
    std::atomic<uint32_t> done{0};
    std::atomic<bool> same{true};
    int helpers = -1;

    void work(const Job j) {
        const uint32_t n = (uint32_t)j.sk->segs.size();
        for (;;) {
            uint64_t cur = next.load(std::memory_order_acquire);
            if ((uint32_t)(cur >> 32) != j.id || (uint32_t)cur >= n) return;
            if (!next.compare_exchange_weak(cur, cur + 1, std::memory_order_acq_rel)) continue;
            const uint32_t i = (uint32_t)cur;
            if (same.load(std::memory_order_relaxed) && !j.sk->match_seg(j.sk->segs[i], j.buf, j.slab))
                same.store(false, std::memory_order_relaxed);
            // (the job cannot end before this increment: the caller waits for done == n)
            if (done.fetch_add(1, std::memory_order_acq_rel) + 1 == n) {
                std::lock_guard<std::mutex> lk(mu);
                cv_done.notify_all();
            }
        }
    }
    void loop() {
        uint32_t seen = 0;
        for (;;) {
            Job j;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_work.wait(lk, [&] { return stop || job.id != seen; });
                if (stop) return;
                j = job;
                seen = j.id;
            }
            work(j);
        }
    }
    bool match(const EvalSkel &s, const uint8_t *b, gc_label *rows) {
        if (helpers < 0) {
            const char *v = std::getenv("GC_STREAM_THREADS");
            helpers = v ? std::atoi(v) : 3;
            helpers = helpers < 0 ? 0 : helpers > 15 ? 15 : helpers;
        }
        if (s.segs.size() < 2 || helpers == 0) {
            for (const EvalSkel::Seg &sg : s.segs)
                if (!s.match_seg(sg, b, rows)) return false;
            return true;
        }
        if (th.empty())
            for (int i = 0; i < helpers; i++) th.emplace_back([this] { loop(); });
        Job j;
        {
            std::lock_guard<std::mutex> lk(mu);
            job.sk = &s, job.buf = b, job.slab = rows;
            job.id++;
            j = job;
            done.store(0, std::memory_order_relaxed);
            same.store(true, std::memory_order_relaxed);
            next.store((uint64_t)j.id << 32, std::memory_order_release);
        }
        cv_work.notify_all();
        work(j);
        const uint32_t n = (uint32_t)s.segs.size();
        if (done.load(std::memory_order_acquire) != n) {
            std::unique_lock<std::mutex> lk(mu);
            cv_done.wait(lk, [&] { return done.load(std::memory_order_acquire) == n; });
        }
        return same.load();
    }
    ~SkelPool() {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
        }
        cv_work.notify_all();
        for (auto &t : th) t.join();
    }
};

}  // namespace gcs
