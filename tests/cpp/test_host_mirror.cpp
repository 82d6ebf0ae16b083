// test_host_mirror.cpp — the reference's own tests for the hot path, re-written against the C++ host
// mirror (include/mpc_host.hpp) that sits on the C ABI.  Needs an MI355X (run from tests/test_gpu_host_mirror.py).
//   TestLabel                 ot/label_test.go:40-92
//   TestGarbleEval            compiler/arithmetic_test.go:102-151 shape: Garble || Eval == plaintext op
//   TestIKNPExpand            ot/iknp_test.go:17-116 (rcvd[i] == sent[i] ^ b[i]*Delta; chunk sizes)
//   TestOT (COT)              ot/ot_test.go:22-127 (alternating flags deliver the chosen label)
//   TestStreaming             circuit/stream_garble.go: same tables as Circuit.Garble for the same R / labels
//   error behaviour           garble.go:260 key size, io.Reader exhaustion, eval.go:54-56 corrupted rows
#include <cstdio>
#include <random>
#include <sstream>

#include "mpc_host.hpp"

using namespace mpc;
using ot::Label;

static int failures = 0;
#define EXPECT(cond, msg)                                            \
    do {                                                             \
        if (!(cond)) {                                               \
            std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, msg); \
            failures++;                                              \
        }                                                            \
    } while (0)

struct PrngReader : Reader {  // deterministic io.Reader (ot_test.go:251-267 style)
    std::mt19937_64 g;
    explicit PrngReader(uint64_t seed) : g(seed) {}
    void Read(uint8_t *buf, size_t n) override {
        for (size_t i = 0; i < n; i++) buf[i] = (uint8_t)g();
    }
};

// test-only base OT: both parties share the object; the 128 base OTs of IKNP are out of scope (CO, EC P-256)
struct ClearBaseOT : ot::OT {
    std::vector<ot::Wire> held;
    void InitSender(ot::IO &) override {}
    void InitReceiver(ot::IO &) override {}
    void Send(const std::vector<ot::Wire> &wires) override { held = wires; }
    void Receive(const std::vector<bool> &flags, std::vector<Label> &result) override {
        for (size_t i = 0; i < flags.size(); i++) result[i] = flags[i] ? held[i].L1 : held[i].L0;
    }
};

static void TestLabel() {
    Label l{0xffffffffffffffffull, 0xffffffffffffffffull};
    l.SetS(true);
    EXPECT(l.D0 == 0xffffffffffffffffull, "Failed to set S-bit");
    l.SetS(false);
    EXPECT(l.D0 == 0x7fffffffffffffffull, "Failed to clear S-bit");
    l = Label{0, 0xffffffffffffffffull};
    l.Mul2();
    EXPECT(l.D0 == 1 && l.D1 == 0xfffffffffffffffeull, "Mul2");
    l = Label{0, 0xffffffffffffffffull};
    l.Mul4();
    EXPECT(l.D0 == 3 && l.D1 == 0xfffffffffffffffcull, "Mul4");
    const uint64_t val = 0x5555555555555555ull;
    l = Label{val, val << 1};
    l.Xor(Label{~0ull, ~0ull});
    EXPECT(l.D0 == (val << 1) && l.D1 == val, "Xor");
}

// n-bit ripple-carry adder as Bristol text: a + b -> n+1 bits, with an OR/INV-based carry so every gate type
// but XNOR is exercised
static std::string AdderBristol(int n) {
    std::ostringstream g;
    int w = 2 * n;  // next free wire
    std::vector<std::string> lines;
    std::vector<int> sum(n + 1);
    int carry = -1;
    for (int i = 0; i < n; i++) {
        int a = i, b = n + i;
        int x = w++;
        lines.push_back("2 1 " + std::to_string(a) + " " + std::to_string(b) + " " + std::to_string(x) + " XOR");
        int ab = w++;
        lines.push_back("2 1 " + std::to_string(a) + " " + std::to_string(b) + " " + std::to_string(ab) + " AND");
        if (carry < 0) {
            sum[i] = x;
            carry = ab;
        } else {
            int s = w++;
            lines.push_back("2 1 " + std::to_string(x) + " " + std::to_string(carry) + " " + std::to_string(s) + " XOR");
            int xc = w++;
            lines.push_back("2 1 " + std::to_string(x) + " " + std::to_string(carry) + " " + std::to_string(xc) + " AND");
            int c2 = w++;
            lines.push_back("2 1 " + std::to_string(ab) + " " + std::to_string(xc) + " " + std::to_string(c2) + " OR");
            sum[i] = s;
            carry = c2;
        }
    }
    // outputs must be the last wires: copy through double inversion / XOR with zero-equivalent
    std::vector<int> outs;
    int base = w + 2 * (n + 1);
    (void)base;
    std::vector<std::string> tail;
    sum[n] = carry;
    for (int i = 0; i <= n; i++) {
        int t = w++;
        tail.push_back("1 1 " + std::to_string(sum[i]) + " " + std::to_string(t) + " INV");
        outs.push_back(t);
    }
    for (int i = 0; i <= n; i++) {
        int t = w++;
        tail.push_back("1 1 " + std::to_string(outs[i]) + " " + std::to_string(t) + " INV");
    }
    g << lines.size() + tail.size() << " " << w << "\n2 " << n << " " << n << "\n1 " << n + 1 << "\n\n";
    for (auto &l : lines) g << l << "\n";
    for (auto &l : tail) g << l << "\n";
    return g.str();
}

static void TestGarbleEval(Context &ctx) {
    const int bits = 3;
    std::istringstream in(AdderBristol(bits));
    circuit::Circuit c = circuit::ParseBristol(in);
    EXPECT(c.InputsSize() == 2 * bits && c.OutputsSize() == bits + 1, "adder shape");
    std::vector<uint8_t> key(32);
    for (int i = 0; i < 32; i++) key[i] = (uint8_t)i;
    PrngReader rand(7);
    for (int gv = 0; gv < (1 << bits); gv++) {
        for (int ev = 0; ev < (1 << bits); ev++) {
            circuit::Garbled g = c.Garble(rand, key, ctx);
            EXPECT(g.R.S(), "R.S()");
            std::vector<Label> wires((size_t)c.NumWires);
            for (int i = 0; i < bits; i++) {
                wires[i] = ((gv >> i) & 1) ? g.Wires[i].L1 : g.Wires[i].L0;
                wires[bits + i] = ((ev >> i) & 1) ? g.Wires[bits + i].L1 : g.Wires[bits + i].L0;
            }
            c.Eval(key, wires, g.Gates, ctx);
            int result = 0;
            for (int i = 0; i <= bits; i++) {  // BitFromLabel (helpers.go:18-28)
                const size_t w = (size_t)c.NumWires - (bits + 1) + i;
                if (wires[w].Equal(g.Wires[w].L1)) result |= 1 << i;
                else EXPECT(wires[w].Equal(g.Wires[w].L0), "unknown label");
            }
            EXPECT(result == gv + ev, "Garbler||Evaluator result != g+e");
            // every wire keeps the free-XOR invariant
            for (auto &w : g.Wires) {
                Label d = w.L0;
                d.Xor(w.L1);
                EXPECT(d.Equal(g.R), "L0^L1 != R");
            }
        }
    }
    // errors
    try {
        std::vector<uint8_t> bad(17);
        c.Garble(rand, bad, ctx);
        EXPECT(false, "expected key size error");
    } catch (const Error &e) {
        EXPECT(std::string(e.what()).find("invalid key size") != std::string::npos, e.what());
    }
    try {
        BytesReader empty({1, 2, 3});
        c.Garble(empty, key, ctx);
        EXPECT(false, "expected EOF");
    } catch (const Error &e) {
        EXPECT(std::string(e.what()) == "unexpected EOF", e.what());
    }
    try {
        circuit::Garbled g = c.Garble(rand, key, ctx);
        std::vector<Label> wires((size_t)c.NumWires);
        auto rows = g.Gates;
        for (size_t i = 0; i < rows.size(); i++)
            if (c.Gates[i].Op == circuit::AND) {
                rows[i].second = 1;
                break;
            }
        c.Eval(key, wires, rows, ctx);
        EXPECT(false, "expected corrupted circuit");
    } catch (const Error &e) {
        EXPECT(std::string(e.what()).find("corrupted ciruit: AND row length: 1") != std::string::npos, e.what());
    }
}

static void expandN(Context &ctx, int n) {
    auto pipe = ot::Pipe::New();
    ClearBaseOT base;
    PrngReader r0(100 + n), r1(200 + n), rb(300 + n);
    std::vector<bool> b((size_t)n);
    for (int i = 0; i < n; i++) {
        uint8_t v;
        rb.Read(&v, 1);
        b[(size_t)i] = v & 1;
    }
    ot::IKNPReceiver rcv(ctx, base, *pipe.second, r1);
    ot::IKNPSender snd(ctx, base, *pipe.first, r0);
    std::vector<Label> rcvd((size_t)n);
    rcv.Receive(b, rcvd);
    std::vector<Label> sent = snd.Send(n);
    for (int i = 0; i < n; i++) {  // iknp_test.go:98-113
        Label x = sent[(size_t)i];
        if (b[(size_t)i]) x.Xor(snd.Delta);
        EXPECT(x.Equal(rcvd[(size_t)i]), "rcvd[i] != sent[i] ^ b[i]*Delta");
    }
}

static void TestIKNPExpand(Context &ctx) {
    expandN(ctx, 129);
    const int cs = (int)ot::chunkSize / 16;
    for (int n : {0, 1, cs, cs + 1, cs * 2, cs * 2 + 1, cs * 3, cs * 3 + 1, cs * 4, cs * 4 + 1, cs * 5}) expandN(ctx, n);
}

static void TestOTCOT(Context &ctx) {
    const int size = 64;
    auto pipe = ot::Pipe::New();
    ClearBaseOT base;
    PrngReader rs(1), rr(2), rw(3);
    ot::COT sender(ctx, base, rs), receiver(ctx, base, rr);
    receiver.InitReceiver(*pipe.second);  // base OTs: the IKNP receiver is the base sender
    sender.InitSender(*pipe.first);
    std::vector<ot::Wire> wires(size);
    std::vector<bool> flags(size);
    for (int i = 0; i < size; i++) {
        wires[(size_t)i].L0 = ot::NewLabel(rw);
        wires[(size_t)i].L1 = ot::NewLabel(rw);
        flags[(size_t)i] = i % 2 == 0;
    }
    std::vector<Label> labels(size);
    receiver.Receive(flags, labels);
    sender.Send(wires);
    receiver.FinishReceive();
    for (int i = 0; i < size; i++) {  // ot_test.go:83-97
        const Label &expected = flags[(size_t)i] ? wires[(size_t)i].L1 : wires[(size_t)i].L0;
        EXPECT(labels[(size_t)i].Equal(expected), "COT delivered the wrong label");
    }
}

static void TestStreaming(Context &ctx) {
    std::istringstream in(AdderBristol(8));
    circuit::Circuit c = circuit::ParseBristol(in);
    std::vector<uint8_t> key(16, 0x42);
    // same byte stream for both: R then one L0 per input wire
    std::vector<uint8_t> rnd(16 * ((size_t)c.InputsSize() + 1));
    PrngReader(99).Read(rnd.data(), rnd.size());
    BytesReader r1(rnd), r2(rnd);
    circuit::Garbled g = c.Garble(r1, key, ctx);
    std::vector<circuit::Wire> inputs, outs;
    for (int i = 0; i < c.InputsSize(); i++) inputs.push_back((circuit::Wire)i);
    for (int i = 0; i < c.OutputsSize(); i++) outs.push_back((circuit::Wire)(1000 + i));
    circuit::Streaming s(r2, key, inputs, ctx);
    std::vector<uint8_t> buf;
    s.Garble(c, inputs, outs, buf);
    // walk the stream: rows of every gate equal Garbled.Gates, in order
    size_t pos = 0;
    for (size_t i = 0; i < c.Gates.size(); i++) {
        uint8_t op = buf[pos++];
        EXPECT((op & 0x0f) == c.Gates[i].Op, "op byte");
        int wc = c.Gates[i].Op == circuit::INV ? 2 : 3;
        pos += (size_t)((op & 0x10) ? 2 : 4) * wc;
        for (size_t r = 0; r < g.Gates[i].second; r++) {
            uint8_t be[16];
            g.Gates[i].first[r].GetData(be);
            EXPECT(std::memcmp(be, buf.data() + pos, 16) == 0, "stream row != Garbled.Gates row");
            pos += 16;
        }
    }
    EXPECT(pos == buf.size(), "stream length");
    for (int i = 0; i < c.OutputsSize(); i++) {
        ot::Wire w = s.GetInput(outs[(size_t)i]);
        EXPECT(w.L0.Equal(g.Wires[(size_t)c.NumWires - c.OutputsSize() + i].L0), "stream output wire");
    }
}

int main() {
    try {
        Context &ctx = Context::Default();
        TestLabel();
        TestGarbleEval(ctx);
        TestIKNPExpand(ctx);
        TestOTCOT(ctx);
        TestStreaming(ctx);
    } catch (const std::exception &e) {
        std::printf("EXCEPTION %s\n", e.what());
        return 2;
    }
    std::printf(failures ? "FAILED (%d)\n" : "ok\n", failures);
    return failures ? 1 : 0;
}
