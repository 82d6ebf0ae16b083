// mpc_host.hpp — C++ host-side mirror of the reference's Go API for the hot path, on top of the C ABI.
//
// The reference is Go; this image has no Go toolchain, so next to the cgo shim source (go/) the host side
// is mirrored here in C++ with the same names, argument meaning and error behaviour:
//
//   mpc::ot::Label / Wire                     ot/label.go:18-166
//   mpc::ot::IO, Pipe                         ot/io.go:15-47, ot/pipe.go:22-135
//   mpc::ot::OT (interface)                   ot/ot.go:16-28
//   mpc::ot::IKNPSender / IKNPReceiver        ot/iknp.go:80-226, 313-511   (semi-honest path)
//   mpc::ot::MITCCRH, COT                     ot/mitccrh.go:50-128, ot/cot.go:51-235
//   mpc::circuit::Operation / Gate / Circuit  circuit/circuit.go:22-34,120-131,260-266
//   Circuit::Garble / Circuit::Eval / Garbled circuit/garble.go:162-183,248-308, circuit/eval.go:17-115
//   mpc::circuit::ParseBristol                circuit/parser.go:265-494
//   mpc::circuit::Streaming                   circuit/stream_garble.go:41-192
//
// Go `error` returns become exceptions of type mpc::Error whose what() carries the reference's message.
// Go io.Reader becomes mpc::Reader.  Everything compute-heavy goes through libgcengine.so; there is no
// CPU fallback.
#pragma once

#include <cstdint>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "gcengine.h"

namespace mpc {

struct Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};

// io.Reader
struct Reader {
    virtual ~Reader() = default;
    virtual void Read(uint8_t *buf, size_t n) = 0;  // throws mpc::Error("unexpected EOF") when exhausted
};
struct BytesReader : Reader {
    std::vector<uint8_t> data;
    size_t pos = 0;
    explicit BytesReader(std::vector<uint8_t> d) : data(std::move(d)) {}
    void Read(uint8_t *buf, size_t n) override {
        if (pos + n > data.size()) throw Error("unexpected EOF");
        std::memcpy(buf, data.data() + pos, n);
        pos += n;
    }
};

inline void check(int st, const char *what) {
    if (st == GC_OK) return;
    std::string msg;
    switch (st) {
    case GC_E_KEYSIZE: msg = "crypto/aes: invalid key size"; break;
    case GC_E_GATE: msg = "invalid gate type"; break;
    case GC_E_ROWS: msg = "corrupted circuit"; break;
    default: msg = std::string(what) + ": " + gc_strerror(st) + " " + gc_last_error();
    }
    throw Error(msg);
}

// one device context per process unless the caller makes more (gc_ctx is one HIP device + stream)
class Context {
public:
    explicit Context(int device = 0) {
        int st = GC_OK;
        h_ = gc_ctx_create(device, &st);
        check(st, "gc_ctx_create");
    }
    ~Context() { gc_ctx_destroy(h_); }
    Context(const Context &) = delete;
    Context &operator=(const Context &) = delete;
    gc_ctx *handle() const { return h_; }
    static Context &Default() {
        static Context c(0);
        return c;
    }

private:
    gc_ctx *h_ = nullptr;
};

namespace ot {

// ot.Label (label.go:28-166).  Same memory layout as gc_label / Go's struct.
struct Label {
    uint64_t D0 = 0, D1 = 0;
    bool Equal(const Label &o) const { return D0 == o.D0 && D1 == o.D1; }
    bool S() const { return (D0 & 0x8000000000000000ull) != 0; }
    void SetS(bool set) {
        if (set) D0 |= 0x8000000000000000ull;
        else D0 &= 0x7fffffffffffffffull;
    }
    void Mul2() {
        D0 <<= 1;
        D0 |= D1 >> 63;
        D1 <<= 1;
    }
    void Mul4() {
        D0 <<= 2;
        D0 |= D1 >> 62;
        D1 <<= 2;
    }
    void Xor(const Label &o) {
        D0 ^= o.D0;
        D1 ^= o.D1;
    }
    unsigned Bit(int i) const {
        if (i < 0 || i > 127) throw Error("invalid bit index");
        return (unsigned)(((i > 63 ? D1 : D0) >> (i & 63)) & 1);
    }
    void GetData(uint8_t buf[16]) const {
        for (int i = 0; i < 8; i++) {
            buf[i] = (uint8_t)(D0 >> (56 - 8 * i));
            buf[8 + i] = (uint8_t)(D1 >> (56 - 8 * i));
        }
    }
    void SetData(const uint8_t buf[16]) {
        D0 = D1 = 0;
        for (int i = 0; i < 8; i++) {
            D0 = (D0 << 8) | buf[i];
            D1 = (D1 << 8) | buf[8 + i];
        }
    }
    std::string String() const {
        char b[40];
        std::snprintf(b, sizeof b, "%016llx%016llx", (unsigned long long)D0, (unsigned long long)D1);
        return b;
    }
};
static_assert(sizeof(Label) == sizeof(gc_label), "Label layout");

inline Label NewLabel(Reader &rand) {  // label.go:46-55
    uint8_t buf[16];
    rand.Read(buf, 16);
    Label l;
    l.SetData(buf);
    return l;
}

struct Wire {
    Label L0, L1;
};
static_assert(sizeof(Wire) == sizeof(gc_wire), "Wire layout");

// ot.IO (io.go:15-47) — the subset the hot path uses
struct IO {
    virtual ~IO() = default;
    virtual void SendData(const std::vector<uint8_t> &d) = 0;
    virtual std::vector<uint8_t> ReceiveData() = 0;
    virtual void Flush() {}
    void SendLabel(const Label &l) {
        std::vector<uint8_t> b(16);
        l.GetData(b.data());
        SendData(b);
    }
    Label ReceiveLabel() {
        auto b = ReceiveData();
        if (b.size() != 16) throw Error("invalid label length");
        Label l;
        l.SetData(b.data());
        return l;
    }
};

// ot.Pipe (pipe.go): in-memory IO for single-threaded, strictly alternating protocols (tests)
class Pipe : public IO {
public:
    static std::pair<std::shared_ptr<Pipe>, std::shared_ptr<Pipe>> New() {
        auto a = std::make_shared<Pipe>(), b = std::make_shared<Pipe>();
        a->peer_ = b.get();
        b->peer_ = a.get();
        return {a, b};
    }
    void SendData(const std::vector<uint8_t> &d) override {
        if (d.size() > 64 * 1024) throw Error("pipe: message too long");  // pipe.go:62-71
        peer_->q_.push_back(d);
    }
    std::vector<uint8_t> ReceiveData() override {
        if (q_.empty()) throw Error("EOF");
        auto d = std::move(q_.front());
        q_.pop_front();
        return d;
    }

private:
    Pipe *peer_ = nullptr;
    std::deque<std::vector<uint8_t>> q_;
};

// ot.OT (ot.go:16-28)
struct OT {
    virtual ~OT() = default;
    virtual void InitSender(IO &io) = 0;
    virtual void InitReceiver(IO &io) = 0;
    virtual void Send(const std::vector<Wire> &wires) = 0;
    virtual void Receive(const std::vector<bool> &flags, std::vector<Label> &result) = 0;
};

constexpr int K = 128;                  // iknp.go:64-67
constexpr size_t chunkSize = 8 * 1024;  // iknp.go:70

// MITCCRH over a run of OTs (mitccrh.go:93-128 as driven by cot.go:160-171)
inline void MITCCRHHash(Context &ctx, const Label &seed, uint64_t gid0, std::vector<Label> &blks, uint32_t h) {
    if (blks.empty()) return;
    check(gc_mitccrh_hash(ctx.handle(), (const gc_label *)&seed, gid0, (gc_label *)blks.data(), blks.size() / h, h),
          "gc_mitccrh_hash");
}

class IKNPSender {
public:
    Label Delta;
    // NewIKNPSender (iknp.go:89-127): k0[i] = the base-OT result for choice Delta.Bit(i)
    IKNPSender(Context &ctx, OT &base, IO &io, Reader &r, const Label *d = nullptr) : ctx_(ctx), io_(io) {
        Delta = d ? *d : NewLabel(r);
        std::vector<bool> flags(K);
        for (int i = 0; i < K; i++) flags[i] = Delta.Bit(i) == 1;
        std::vector<Label> k0(K);
        base.Receive(flags, k0);
        int st = GC_OK;
        h_ = gc_iknp_sender_create(ctx.handle(), (const gc_label *)&Delta, (const gc_label *)k0.data(), &st);
        check(st, "gc_iknp_sender_create");
    }
    ~IKNPSender() { gc_iknp_free(h_); }
    // Send (iknp.go:129, semi-honest): returns b0; b1 = b0 ^ Delta
    std::vector<Label> Send(int n) {
        std::vector<Label> result((size_t)n);
        const size_t want = gc_iknp_u_bytes((size_t)n);
        std::vector<uint8_t> u;
        while (u.size() < want) {  // send(): one ReceiveData per chunk (iknp.go:202-209)
            auto chunk = io_.ReceiveData();
            if (chunk.size() % K != 0) throw Error("invalid chunk size: " + std::to_string(chunk.size()));
            u.insert(u.end(), chunk.begin(), chunk.end());
        }
        if (n) check(gc_iknp_send(h_, u.data(), u.size(), (size_t)n, (gc_label *)result.data()), "gc_iknp_send");
        return result;
    }

private:
    Context &ctx_;
    IO &io_;
    gc_iknp *h_ = nullptr;
};

class IKNPReceiver {
public:
    // NewIKNPReceiver (iknp.go:321-361)
    IKNPReceiver(Context &ctx, OT &base, IO &io, Reader &rand) : ctx_(ctx), io_(io) {
        std::vector<Wire> wires(K);
        for (int i = 0; i < K; i++) {
            wires[i].L0 = NewLabel(rand);
            wires[i].L1 = NewLabel(rand);
        }
        base.Send(wires);
        int st = GC_OK;
        h_ = gc_iknp_receiver_create(ctx.handle(), (const gc_wire *)wires.data(), &st);
        check(st, "gc_iknp_receiver_create");
    }
    ~IKNPReceiver() { gc_iknp_free(h_); }
    // Receive (iknp.go:364, semi-honest): result[i] = b0[i] ^ b[i]*Delta
    void Receive(const std::vector<bool> &b, std::vector<Label> &result) {
        if (b.size() != result.size()) throw Error("len(b) != len(result)");
        const size_t n = b.size();
        if (n) {
            std::vector<uint8_t> choice(n), u(gc_iknp_u_bytes(n));
            for (size_t i = 0; i < n; i++) choice[i] = b[i] ? 1 : 0;
            check(gc_iknp_receive(h_, choice.data(), n, u.data(), (gc_label *)result.data()), "gc_iknp_receive");
            for (size_t ofs = 0; ofs < u.size(); ofs += chunkSize)  // SendData per chunk (iknp.go:499)
                io_.SendData(std::vector<uint8_t>(u.begin() + (long)ofs,
                                                  u.begin() + (long)std::min(u.size(), ofs + chunkSize)));
        }
        io_.Flush();
    }

private:
    Context &ctx_;
    IO &io_;
    gc_iknp *h_ = nullptr;
};

// ot.COT (cot.go:51-235), semi-honest
class COT : public OT {
public:
    COT(Context &ctx, OT &base, Reader &r) : ctx_(ctx), base_(base), r_(r) {}
    void InitSender(IO &io) override {
        if (iknpR_) throw Error("already initialized as receiver");
        if (iknpS_) throw Error("already initialized");
        base_.InitSender(io);
        iknpS_ = std::make_unique<IKNPSender>(ctx_, base_, io, r_);
        io_ = &io;
    }
    void InitReceiver(IO &io) override {
        if (iknpS_) throw Error("already initialized as sender");
        if (iknpR_) throw Error("already initialized");
        base_.InitReceiver(io);
        iknpR_ = std::make_unique<IKNPReceiver>(ctx_, base_, io, r_);
        io_ = &io;
    }
    void Send(const std::vector<Wire> &wires) override {  // cot.go:136-185
        if (!iknpS_) throw Error("not initialized as sender");
        auto data = iknpS_->Send((int)wires.size());
        Label seed = NewLabel(r_);
        io_->SendLabel(seed);
        io_->Flush();
        std::vector<Label> out(2 * wires.size());
        if (!wires.empty())
            check(gc_cot_send_pads(ctx_.handle(), (const gc_label *)&seed, (const gc_label *)&iknpS_->Delta,
                                   (const gc_label *)data.data(), (const gc_wire *)wires.data(), wires.size(),
                                   (gc_label *)out.data()),
                  "gc_cot_send_pads");
        for (const Label &l : out) io_->SendLabel(l);
        io_->Flush();
    }
    void Receive(const std::vector<bool> &flags, std::vector<Label> &result) override {  // cot.go:187-235
        if (!iknpR_) throw Error("not initialized as receiver");
        iknpR_->Receive(flags, result);
        pending_flags_ = flags;
        pending_ = &result;
    }
    // second half of COT.Receive once the sender's messages are in the pipe (single-threaded tests run
    // the two parties in lock-step; a threaded driver calls Receive() then FinishReceive() back to back)
    void FinishReceive() {
        std::vector<Label> &result = *pending_;
        Label seed = io_->ReceiveLabel();
        std::vector<Label> sent(2 * result.size());
        for (auto &l : sent) l = io_->ReceiveLabel();
        std::vector<uint8_t> f(result.size());
        for (size_t i = 0; i < f.size(); i++) f[i] = pending_flags_[i] ? 1 : 0;
        if (!result.empty())
            check(gc_cot_receive_unpad(ctx_.handle(), (const gc_label *)&seed, f.data(), (const gc_label *)sent.data(),
                                       (gc_label *)result.data(), result.size()),
                  "gc_cot_receive_unpad");
    }
    IKNPSender *sender() { return iknpS_.get(); }

private:
    Context &ctx_;
    OT &base_;
    Reader &r_;
    IO *io_ = nullptr;
    std::unique_ptr<IKNPSender> iknpS_;
    std::unique_ptr<IKNPReceiver> iknpR_;
    std::vector<bool> pending_flags_;
    std::vector<Label> *pending_ = nullptr;
};

}  // namespace ot

namespace circuit {

// circuit.Operation (circuit.go:25-34)
enum Operation : uint8_t { XOR = 0, XNOR = 1, AND = 2, OR = 3, INV = 4 };
inline const char *OperationString(Operation op) {
    static const char *n[] = {"XOR", "XNOR", "AND", "OR", "INV"};
    return op <= INV ? n[op] : "{Operation}";
}

using Wire = uint32_t;

// circuit.Gate (circuit.go:260-266), 20 bytes like Go's (circuit_test.go:14-19)
struct Gate {
    Wire Input0, Input1, Output;
    Operation Op;
    uint32_t Level;
};
static_assert(sizeof(Gate) == 20 && sizeof(Gate) == sizeof(gc_gate), "Gate layout");

class Circuit;

// circuit.Garbled (garble.go:162-183)
struct Garbled {
    ot::Label R;
    std::vector<ot::Wire> Wires;
    std::vector<std::pair<const ot::Label *, size_t>> Gates;  // (rows, count) per gate; (nullptr,0) for free gates
    std::vector<ot::Label> slab;                             // backing store of Gates
    void Release() {}                                        // scratch is owned by value here
};

class Circuit {
public:
    int NumGates = 0, NumWires = 0;
    std::vector<int> Inputs, Outputs;  // IO sizes in bits (IO.Size() = sum)
    std::vector<Gate> Gates;

    int InputsSize() const { return sum(Inputs); }
    int OutputsSize() const { return sum(Outputs); }

    // Garble (garble.go:248-308)
    Garbled Garble(Reader &rand, const std::vector<uint8_t> &key, Context &ctx = Context::Default()) {
        const int nin = InputsSize();
        std::vector<uint8_t> rnd(16 * ((size_t)nin + 1));
        rand.Read(rnd.data(), 16);                               // R             (:253)
        if (key.size() != 16 && key.size() != 24 && key.size() != 32)
            throw Error("crypto/aes: invalid key size " + std::to_string(key.size()));  // (:260)
        rand.Read(rnd.data() + 16, 16 * (size_t)nin);             // input labels  (:271-278)
        gc_circ *c = device(ctx);
        gc_plan_info info;
        gc_plan_get_info(gc_circ_plan(c), &info);
        Garbled g;
        g.Wires.resize((size_t)NumWires);
        g.slab.resize(info.slab_rows);
        check(gc_garble(c, key.data(), key.size(), rnd.data(), rnd.size(), 1, (gc_label *)&g.R,
                        (gc_wire *)g.Wires.data(), nullptr, (gc_label *)g.slab.data()),
              "gc_garble");
        g.Gates.resize(Gates.size());
        size_t off = 0;
        for (size_t i = 0; i < Gates.size(); i++) {
            size_t n = Gates[i].Op == AND ? 2 : Gates[i].Op == OR ? 3 : Gates[i].Op == INV ? 1 : 0;
            g.Gates[i] = {n ? g.slab.data() + off : nullptr, n};
            off += n;
        }
        return g;
    }

    // Eval (eval.go:17-115): wires (len NumWires, inputs pre-filled) is written in place
    void Eval(const std::vector<uint8_t> &key, std::vector<ot::Label> &wires,
              const std::vector<std::pair<const ot::Label *, size_t>> &garbled, Context &ctx = Context::Default()) {
        if (key.size() != 16 && key.size() != 24 && key.size() != 32)
            throw Error("crypto/aes: invalid key size " + std::to_string(key.size()));
        std::vector<ot::Label> slab;
        for (size_t i = 0; i < Gates.size(); i++) {
            const auto &row = garbled[i];
            switch (Gates[i].Op) {
            case AND:
                if (row.second != 2) throw Error("corrupted ciruit: AND row length: " + std::to_string(row.second));
                break;
            case OR:
                if (row.second < 3) throw Error("corrupted circuit: index " + std::to_string(row.second) + " >= row " + std::to_string(row.second));
                break;
            case INV:
                if (row.second < 1) throw Error("corrupted circuit: index 0 >= row 0");
                break;
            default: continue;
            }
            slab.insert(slab.end(), row.first, row.first + row.second);
        }
        gc_circ *c = device(ctx);
        check(gc_eval(c, key.data(), key.size(), 1, (gc_label *)wires.data(), nullptr, (const gc_label *)slab.data(),
                      slab.size(), nullptr),
              "gc_eval");
    }

    ~Circuit() {
        if (circ_) gc_circ_free(circ_);
    }
    Circuit() = default;
    Circuit(Circuit &&o) noexcept { *this = std::move(o); }
    Circuit &operator=(Circuit &&o) noexcept {
        NumGates = o.NumGates;
        NumWires = o.NumWires;
        Inputs = std::move(o.Inputs);
        Outputs = std::move(o.Outputs);
        Gates = std::move(o.Gates);
        circ_ = o.circ_;
        o.circ_ = nullptr;
        return *this;
    }

    gc_circ *device(Context &ctx) {
        std::lock_guard<std::mutex> lk(mu_);
        if (!circ_) {
            int st = GC_OK;
            circ_ = gc_circ_load(ctx.handle(), (const gc_gate *)Gates.data(), (uint32_t)Gates.size(), (uint32_t)NumWires,
                                 (uint32_t)InputsSize(), (uint32_t)OutputsSize(), &st);
            if (st == GC_E_GATE) throw Error("invalid gate type");
            check(st, "gc_circ_load");
        }
        return circ_;
    }

private:
    static int sum(const std::vector<int> &v) {
        int s = 0;
        for (int x : v) s += x;
        return s;
    }
    gc_circ *circ_ = nullptr;
    std::mutex mu_;
};

// ParseBristol (parser.go:265-494), same validation messages
inline Circuit ParseBristol(std::istream &in) {
    auto fail = [](const std::string &m) -> void { throw Error(m); };
    std::string line;
    auto next = [&](std::vector<std::string> &parts) {
        while (std::getline(in, line)) {
            std::istringstream ss(line);
            parts.clear();
            std::string t;
            while (ss >> t) parts.push_back(t);
            if (!parts.empty()) return true;
        }
        return false;
    };
    std::vector<std::string> p;
    Circuit c;
    if (!next(p) || p.size() != 2) fail("invalid 1st line: '" + line + "'");
    c.NumGates = std::stoi(p[0]);
    c.NumWires = std::stoi(p[1]);
    std::vector<bool> seen((size_t)c.NumWires, false);
    if (!next(p)) fail("EOF");
    if ((size_t)std::stoi(p[0]) + 1 != p.size()) fail("invalid inputs line: niv=" + p[0] + ", len=" + std::to_string(p.size()));
    long iw = 0;
    for (size_t i = 1; i < p.size(); i++) {
        c.Inputs.push_back(std::stoi(p[i]));
        iw += c.Inputs.back();
    }
    if (iw == 0) fail("no inputs defined");
    for (long i = 0; i < iw; i++) seen.at((size_t)i) = true;
    if (!next(p) || (size_t)std::stoi(p[0]) + 1 != p.size()) fail("invalid outputs line");
    for (size_t i = 1; i < p.size(); i++) c.Outputs.push_back(std::stoi(p[i]));
    int gate = 0;
    for (; next(p); gate++) {
        if (gate >= c.NumGates) fail("too many gates");
        if (p.size() < 3) fail("invalid gate: " + line);
        int n1 = std::stoi(p[0]), n2 = std::stoi(p[1]);
        if ((size_t)(2 + n1 + n2 + 1) != p.size()) fail("invalid gate: " + line);
        std::vector<Wire> ins, outs;
        for (int i = 0; i < n1; i++) {
            unsigned long v = std::stoul(p[2 + i]);
            if (v >= seen.size()) fail("invalid wire " + std::to_string(v) + " [0..." + std::to_string(seen.size()) + "[");
            if (!seen[v]) fail("input " + std::to_string(v) + " of gate " + std::to_string(gate) + " not set");
            ins.push_back((Wire)v);
        }
        for (int i = 0; i < n2; i++) {
            unsigned long v = std::stoul(p[2 + n1 + i]);
            if (v >= seen.size()) fail("invalid wire " + std::to_string(v) + " [0..." + std::to_string(seen.size()) + "[");
            seen[v] = true;
            outs.push_back((Wire)v);
        }
        const std::string &o = p.back();
        Operation op;
        size_t want = 2;
        if (o == "XOR") op = XOR;
        else if (o == "XNOR") op = XNOR;
        else if (o == "AND") op = AND;
        else if (o == "OR") op = OR;
        else if (o == "INV") {
            op = INV;
            want = 1;
        } else {
            fail("invalid operation '" + o + "'");
            op = XOR;
        }
        if (ins.size() != want) fail("invalid number of inputs " + std::to_string(ins.size()) + " for " + o);
        if (outs.size() != 1) fail("invalid number of outputs " + std::to_string(outs.size()) + " for " + o);
        c.Gates.push_back(Gate{ins[0], ins.size() > 1 ? ins[1] : 0, outs[0], op, 0});
    }
    if (gate != c.NumGates) fail("not enough gates: got " + std::to_string(gate) + ", expected " + std::to_string(c.NumGates));
    for (size_t i = 0; i < seen.size(); i++)
        if (!seen[i]) fail("wire " + std::to_string(i) + " not assigned");
    return c;
}

// circuit.Streaming (stream_garble.go:27-192); conn.WriteBuf becomes the caller's byte vector
class Streaming {
public:
    Streaming(Reader &rand, const std::vector<uint8_t> &key, const std::vector<Wire> &inputs,
              Context &ctx = Context::Default()) {
        std::vector<uint8_t> rnd(16 * (inputs.size() + 1));
        rand.Read(rnd.data(), 16);
        if (key.size() != 16 && key.size() != 24 && key.size() != 32)
            throw Error("crypto/aes: invalid key size " + std::to_string(key.size()));
        rand.Read(rnd.data() + 16, 16 * inputs.size());
        int st = GC_OK;
        h_ = gc_stream_create(ctx.handle(), key.data(), key.size(), rnd.data(), rnd.size(), inputs.data(),
                              (uint32_t)inputs.size(), &st);
        check(st, "gc_stream_create");
    }
    ~Streaming() { gc_stream_free(h_); }
    ot::Wire GetInput(Wire w) {
        ot::Wire out;
        check(gc_stream_get_wire(h_, w, (gc_wire *)&out), "gc_stream_get_wire");
        return out;
    }
    // Garble (stream_garble.go:161): appends the serialised gates to buf (conn.WriteBuf)
    void Garble(Circuit &c, const std::vector<Wire> &in, const std::vector<Wire> &out, std::vector<uint8_t> &buf) {
        size_t need = 0;
        const size_t at = buf.size();
        buf.resize(at + c.Gates.size() * 61 + 16);
        check(gc_stream_garble(h_, (const gc_gate *)c.Gates.data(), (uint32_t)c.Gates.size(), (uint32_t)c.NumWires,
                               in.data(), (uint32_t)in.size(), out.data(), (uint32_t)out.size(), buf.data() + at,
                               buf.size() - at, &need),
              "gc_stream_garble");
        buf.resize(at + need);
    }

private:
    gc_stream *h_ = nullptr;
};

}  // namespace circuit
}  // namespace mpc
