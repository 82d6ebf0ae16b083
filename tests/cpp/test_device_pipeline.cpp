// test_device_pipeline.cpp — BASELINE config 2 (aes_128 x 1 024 instances) device-resident, driven through the C ABI
// ALONE: no torch, no HIP header, no Python in this process — the calls a Go host makes through cgo
// (go/circuit/batch_hip.go), in the same order:
//   gc_ctx_create -> gc_circ_load -> gc_batch_create x2 -> gc_dev_alloc / gc_dev_upload (random streams, input bits)
//   -> gc_batch_garble -> gc_batch_select_inputs -> gc_batch_eval -> gc_batch_decode -> gc_dev_download
// and the same five kernels recorded once with gc_ctx_capture_* and replayed.
// Checks (the oracle is the CHECKER only — liboracle.so, tests/ may link it):
//   * decoded output bits of EVERY instance == plaintext evaluation of the circuit (circuit/computer.go:15-91);
//   * decode mismatch counter == 0 (every evaluated output label is one of the garbler's two labels);
//   * R, every table row and the output-wire labels of sampled instances == the oracle's restatement of
//     Circuit.Garble (circuit/garble.go:248-482) byte for byte, evaluated output labels == oracle Eval (eval.go:17-115).
// Replaces the per-call loop of circuit/garble.go:285-299 / eval.go:37-112 for a batch of instances.
// usage: test_device_pipeline <gates.bin> [batch]      gates.bin: u32 ngates,nwires,ninputs,noutputs then gc_gate[ngates]
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "gcengine.h"
#include "../../oracle/oracle.h"

static int failures = 0;
#define EXPECT(cond, ...)                                     \
    do {                                                      \
        if (!(cond)) {                                        \
            std::printf("FAIL %s:%d ", __FILE__, __LINE__);   \
            std::printf(__VA_ARGS__);                         \
            std::printf("\n");                                \
            if (++failures > 20) std::exit(1);                \
        }                                                     \
    } while (0)
#define OK(call)                                                                                   \
    do {                                                                                           \
        int rc__ = (call);                                                                         \
        if (rc__ != GC_OK) {                                                                       \
            std::printf("FAIL %s:%d %s -> %s [%s]\n", __FILE__, __LINE__, #call, gc_strerror(rc__), \
                        gc_last_error());                                                          \
            std::exit(1);                                                                          \
        }                                                                                          \
    } while (0)

int main(int argc, char **argv) {
    if (argc < 2) {
        std::printf("usage: %s gates.bin [batch]\n", argv[0]);
        return 2;
    }
    static_assert(sizeof(gc_gate) == 20 && sizeof(orc_gate) == 20, "circuit.Gate is 20 bytes");
    std::FILE *f = std::fopen(argv[1], "rb");
    if (!f) return 2;
    uint32_t hdr[4];
    if (std::fread(hdr, 4, 4, f) != 4) return 2;
    const uint32_t ngates = hdr[0], nwires = hdr[1], nin = hdr[2], nout = hdr[3];
    std::vector<gc_gate> gates(ngates);
    if (std::fread(gates.data(), sizeof(gc_gate), ngates, f) != ngates) return 2;
    std::fclose(f);
    const uint32_t batch = argc > 2 ? (uint32_t)std::atoi(argv[2]) : 1024;

    uint8_t key[32];
    for (int i = 0; i < 32; i++) key[i] = (uint8_t)i;  // AES-256, as circuit.Garbler uses (garbler.go:47-51)
    std::mt19937_64 rng(20260928);
    const size_t stride = 16 * ((size_t)nin + 1);
    std::vector<uint8_t> rnd(stride * batch), bits((size_t)batch * nin);
    for (auto &b : rnd) b = (uint8_t)rng();
    for (auto &b : bits) b = (uint8_t)(rng() & 1);

    if (gc_abi_version() != GC_ABI_VERSION) {
        std::printf("FAIL ABI %d != %d\n", gc_abi_version(), GC_ABI_VERSION);
        return 1;
    }
    int st = 0;
    gc_ctx *ctx = gc_ctx_create(0, &st);
    if (!ctx) {
        std::printf("FAIL gc_ctx_create: %s [%s]\n", gc_strerror(st), gc_last_error());
        return 1;
    }
    gc_circ *circ = gc_circ_load(ctx, gates.data(), ngates, nwires, nin, nout, &st);
    OK(st);
    gc_plan_info info;
    OK(gc_plan_get_info(gc_circ_plan(circ), &info));
    gc_batch *gb = gc_batch_create(circ, batch, &st);
    OK(st);
    gc_batch *ev = gc_batch_create(circ, batch, &st);
    OK(st);

    // device buffers through the ABI: what a host without a HIP allocator does
    void *d_rnd = gc_dev_alloc(ctx, rnd.size(), &st);
    OK(st);
    void *d_bits = gc_dev_alloc(ctx, bits.size(), &st);
    OK(st);
    void *d_out = gc_dev_alloc(ctx, (size_t)batch * nout, &st);
    OK(st);
    void *d_mis = gc_dev_alloc(ctx, 4, &st);
    OK(st);
    OK(gc_dev_upload(ctx, d_rnd, rnd.data(), rnd.size()));
    OK(gc_dev_upload(ctx, d_bits, bits.data(), bits.size()));
    OK(gc_dev_memset(ctx, d_out, 0xee, (size_t)batch * nout));
    OK(gc_dev_memset(ctx, d_mis, 0, 4));

    auto step = [&] {
        OK(gc_batch_garble(gb, key, sizeof key, d_rnd));
        OK(gc_batch_select_inputs(ev, gb, d_bits));
        OK(gc_batch_eval(ev, key, sizeof key, gb));
        OK(gc_batch_decode(gb, ev, d_out, d_mis));
    };
    step();
    std::vector<uint8_t> out((size_t)batch * nout);
    uint32_t mis = 1;
    OK(gc_dev_download(ctx, out.data(), d_out, out.size()));
    OK(gc_dev_download(ctx, &mis, d_mis, 4));
    EXPECT(mis == 0, "decode mismatches: %u", mis);

    // every instance against plaintext evaluation
    std::vector<uint8_t> wb(nwires);
    for (uint32_t i = 0; i < batch; i++) {
        EXPECT(orc_compute((const orc_gate *)gates.data(), ngates, nwires, nin, &bits[(size_t)i * nin], wb.data()) == 0,
               "orc_compute");
        EXPECT(std::memcmp(&wb[nwires - nout], &out[(size_t)i * nout], nout) == 0, "instance %u: decoded bits != plaintext", i);
    }

    // sampled instances byte for byte against the oracle's Garble / Eval
    std::vector<gc_label> R(batch), slab((size_t)batch * info.slab_rows), gout((size_t)batch * nout),
        eout((size_t)batch * nout);
    OK(gc_batch_read_r(gb, R.data()));
    OK(gc_batch_read_slab(gb, slab.data()));
    OK(gc_batch_read_outputs(gb, gout.data()));
    OK(gc_batch_read_outputs(ev, eout.data()));
    std::vector<orc_wire> ow(nwires);
    std::vector<orc_label> oslab(info.slab_rows + 1), el(nwires);
    const uint32_t samples[] = {0, 1, 3, 4, 63, 64, batch / 2, batch - 2, batch - 1};
    for (uint32_t i : samples) {
        if (i >= batch) continue;
        orc_label r;
        long rows = orc_garble((const orc_gate *)gates.data(), ngates, nwires, nin, key, sizeof key, &rnd[i * stride], stride,
                               &r, ow.data(), oslab.data(), oslab.size(), nullptr);
        EXPECT(rows == (long)info.slab_rows, "oracle rows %ld != %u", rows, info.slab_rows);
        EXPECT(std::memcmp(&r, &R[i], 16) == 0, "instance %u: R differs", i);
        EXPECT(std::memcmp(oslab.data(), &slab[(size_t)i * info.slab_rows], 16 * (size_t)info.slab_rows) == 0,
               "instance %u: garbled tables differ", i);
        for (uint32_t k = 0; k < nout; k++)
            EXPECT(std::memcmp(&ow[nwires - nout + k].l0, &gout[(size_t)i * nout + k], 16) == 0,
                   "instance %u: output wire %u L0 differs", i, k);
        for (uint32_t w = 0; w < nin; w++) el[w] = bits[(size_t)i * nin + w] ? ow[w].l1 : ow[w].l0;
        EXPECT(orc_eval((const orc_gate *)gates.data(), ngates, nwires, key, sizeof key, el.data(), oslab.data(),
                        (size_t)rows) == 0, "orc_eval");
        EXPECT(std::memcmp(&el[nwires - nout], &eout[(size_t)i * nout], 16 * (size_t)nout) == 0,
               "instance %u: evaluated output labels differ", i);
    }

    // the same pipeline as ONE graph launch per step, on fresh inputs in the same buffers
    gc_graph *g = nullptr;
    OK(gc_ctx_capture_begin(ctx));
    step();
    OK(gc_ctx_capture_end(ctx, &g));
    for (auto &b : rnd) b = (uint8_t)rng();
    for (auto &b : bits) b = (uint8_t)(rng() & 1);
    OK(gc_dev_upload(ctx, d_rnd, rnd.data(), rnd.size()));
    OK(gc_dev_upload(ctx, d_bits, bits.data(), bits.size()));
    for (int rep = 0; rep < 3; rep++) OK(gc_graph_launch(g));
    OK(gc_dev_download(ctx, out.data(), d_out, out.size()));
    OK(gc_dev_download(ctx, &mis, d_mis, 4));
    EXPECT(mis == 0, "decode mismatches after graph replay: %u", mis);
    for (uint32_t i = 0; i < batch; i += 7) {
        orc_compute((const orc_gate *)gates.data(), ngates, nwires, nin, &bits[(size_t)i * nin], wb.data());
        EXPECT(std::memcmp(&wb[nwires - nout], &out[(size_t)i * nout], nout) == 0, "graph replay, instance %u", i);
    }
    OK(gc_batch_read_slab(gb, slab.data()));
    {
        orc_label r;
        const uint32_t i = batch - 1;
        orc_garble((const orc_gate *)gates.data(), ngates, nwires, nin, key, sizeof key, &rnd[i * stride], stride, &r,
                   ow.data(), oslab.data(), oslab.size(), nullptr);
        EXPECT(std::memcmp(oslab.data(), &slab[(size_t)i * info.slab_rows], 16 * (size_t)info.slab_rows) == 0,
               "graph replay: garbled tables of the last instance differ");
    }
    // argument errors of the new calls
    EXPECT(gc_dev_upload(ctx, nullptr, rnd.data(), 16) == GC_E_ARG, "upload to NULL");
    EXPECT(gc_dev_download(nullptr, out.data(), d_out, 16) == GC_E_ARG, "download without ctx");
    EXPECT(gc_dev_alloc(nullptr, 16, &st) == nullptr && st == GC_E_ARG, "alloc without ctx");

    gc_graph_free(g);
    gc_dev_free(ctx, d_rnd);
    gc_dev_free(ctx, d_bits);
    gc_dev_free(ctx, d_out);
    gc_dev_free(ctx, d_mis);
    gc_batch_free(gb);
    gc_batch_free(ev);
    gc_circ_free(circ);
    gc_ctx_destroy(ctx);
    if (failures) {
        std::printf("%d failure(s)\n", failures);
        return 1;
    }
    std::printf("aes-shaped circuit: %u gates x %u instances device-resident through the C ABI\nok\n", ngates, batch);
    return 0;
}
