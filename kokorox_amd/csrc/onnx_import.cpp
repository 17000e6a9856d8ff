// `model.onnx` -> KXHIPW01 weight image (see onnx_import.h).  Plain C++17, no HIP: also built with g++ and the
// address / undefined-behaviour sanitizers by the CPU suite's fuzz test.
//
// Field numbers follow the public onnx.proto3 schema:
//   ModelProto      7 graph
//   GraphProto      1 node*, 2 name, 5 initializer*
//   NodeProto       1 input*, 2 output*, 3 name, 4 op_type, 5 attribute*, 7 domain
//   AttributeProto  1 name, 2 f, 3 i, 4 s, 5 t, 7 floats*, 8 ints*
//   TensorProto     1 dims*, 2 data_type, 4 float_data*, 5 int32_data*, 7 int64_data*, 8 name, 9 raw_data,
//                   10 double_data*, 11 uint64_data*, 13 external_data*, 14 data_location
// Every read is bounds-checked against the enclosing message; nothing is allocated from a size the file merely CLAIMS
// (element counts are checked against the payload that is actually there before any buffer is sized from them).
#include "onnx_import.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <memory>
#include <set>

namespace kx {
namespace {

[[noreturn]] void fail(const std::string& m) { throw ImportError(m); }

struct Span {
    const uint8_t* p = nullptr;
    size_t n = 0;
};

std::string str(Span s) { return std::string(reinterpret_cast<const char*>(s.p), s.n); }

bool ends_with(const std::string& s, const std::string& suf) {
    return s.size() >= suf.size() && s.compare(s.size() - suf.size(), suf.size(), suf) == 0;
}
bool starts_with(const std::string& s, const std::string& pre) { return s.compare(0, pre.size(), pre) == 0; }

// ---- protobuf wire format ---------------------------------------------------------------------------------------
struct Field {
    uint64_t no = 0;
    int wt = 0;
    uint64_t v = 0;  // wire type 0
    Span s;          // wire types 1 (8 bytes), 2 (payload), 5 (4 bytes)
};

struct Msg {
    const uint8_t* p;
    const uint8_t* end;
    explicit Msg(Span s) : p(s.p), end(s.p + s.n) {}
    uint64_t varint() {
        uint64_t out = 0;
        for (int shift = 0; shift < 64; shift += 7) {
            if (p >= end) fail("truncated varint");
            const uint8_t b = *p++;
            out |= (uint64_t)(b & 0x7F) << shift;
            if (b < 0x80) return out;
        }
        fail("malformed varint");
    }
    bool next(Field& f) {
        if (p >= end) return false;
        const uint64_t key = varint();
        f.no = key >> 3;
        f.wt = (int)(key & 7);
        const size_t left = (size_t)(end - p);
        switch (f.wt) {
            case 0: f.v = varint(); break;
            case 1:
                if (left < 8) fail("truncated 64-bit field");
                f.s = Span{p, 8};
                p += 8;
                break;
            case 2: {
                const uint64_t len = varint();
                if (len > (uint64_t)(end - p)) fail("truncated length-delimited field");
                f.s = Span{p, (size_t)len};
                p += len;
                break;
            }
            case 5:
                if (left < 4) fail("truncated 32-bit field");
                f.s = Span{p, 4};
                p += 4;
                break;
            default: fail("unsupported protobuf wire type " + std::to_string(f.wt));
        }
        return true;
    }
};

template <class Tp>
Tp load(const uint8_t* p) {
    Tp v;
    memcpy(&v, p, sizeof(Tp));
    return v;
}

void packed_varints(Span s, std::vector<int64_t>& out) {
    Msg m(s);
    while (m.p < m.end) out.push_back((int64_t)m.varint());
}

// repeated scalar: packed (wire type 2) or one element of wire type `single_wt`
void repeated_ints(const Field& f, std::vector<int64_t>& out, const char* what) {
    if (f.wt == 2) packed_varints(f.s, out);
    else if (f.wt == 0) out.push_back((int64_t)f.v);
    else fail(std::string("bad encoding of ") + what);
}
void repeated_f32(const Field& f, std::vector<float>& out, const char* what) {
    if (f.wt == 2) {
        if (f.s.n % 4) fail(std::string("bad packed length of ") + what);
        for (size_t i = 0; i < f.s.n; i += 4) out.push_back(load<float>(f.s.p + i));
    } else if (f.wt == 5) out.push_back(load<float>(f.s.p));
    else fail(std::string("bad encoding of ") + what);
}
void repeated_f64(const Field& f, std::vector<double>& out, const char* what) {
    if (f.wt == 2) {
        if (f.s.n % 8) fail(std::string("bad packed length of ") + what);
        for (size_t i = 0; i < f.s.n; i += 8) out.push_back(load<double>(f.s.p + i));
    } else if (f.wt == 1) out.push_back(load<double>(f.s.p));
    else fail(std::string("bad encoding of ") + what);
}

// ---- messages -----------------------------------------------------------------------------------------------------
enum DType { FLOAT = 1, UINT8, INT8, UINT16, INT16, INT32, INT64, STRING, BOOL, FLOAT16, DOUBLE, UINT32, UINT64, BFLOAT16 = 16 };

int item_size(int dt) {
    switch (dt) {
        case FLOAT: case INT32: case UINT32: return 4;
        case UINT8: case INT8: case BOOL: return 1;
        case UINT16: case INT16: case FLOAT16: case BFLOAT16: return 2;
        case INT64: case DOUBLE: case UINT64: return 8;
        default: return 0;
    }
}

float half_to_float(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000) << 16;
    uint32_t exp = (h >> 10) & 0x1F, man = h & 0x3FF, bits;
    if (exp == 0) {
        if (man == 0) bits = sign;
        else {  // subnormal: normalise
            int e = -1;
            do { man <<= 1; ++e; } while (!(man & 0x400));
            bits = sign | (uint32_t)(127 - 15 - e) << 23 | (man & 0x3FF) << 13;
        }
    } else if (exp == 31) bits = sign | 0x7F800000u | man << 13;
    else bits = sign | (exp + 127 - 15) << 23 | man << 13;
    float f;
    memcpy(&f, &bits, 4);
    return f;
}
float bf16_to_float(uint16_t h) {
    const uint32_t bits = (uint32_t)h << 16;
    float f;
    memcpy(&f, &bits, 4);
    return f;
}

struct Tensor {
    std::string name;
    std::vector<int64_t> dims;
    int dtype = 0;
    bool has_raw = false;
    Span raw;
    std::vector<float> floats;
    std::vector<int64_t> ints;
    std::vector<double> doubles;
    size_t count = 0;
};

using Shape = std::vector<int64_t>;
struct Arr {
    Shape shape;
    std::vector<float> v;
    size_t size() const { return v.size(); }
};

size_t shape_count(const Shape& s, const std::string& who) {
    const size_t cap = (size_t)1 << 40;  // far above any real tensor; keeps every later product inside 64 bits
    size_t c = 1;
    for (int64_t d : s) {
        if (d < 0) fail("tensor " + who + ": negative dimension");
        if (d != 0 && c > cap / (size_t)d) fail("tensor " + who + ": dimensions overflow");
        c *= (size_t)d;
    }
    return c;
}

std::string shape_str(const Shape& s) {
    std::string o = "(";
    for (size_t i = 0; i < s.size(); ++i) o += (i ? ", " : "") + std::to_string(s[i]);
    if (s.size() == 1) o += ",";
    return o + ")";
}

Tensor parse_tensor(Span buf) {
    Tensor t;
    bool external = false;
    Msg m(buf);
    Field f;
    while (m.next(f)) {
        switch (f.no) {
            case 1: repeated_ints(f, t.dims, "TensorProto.dims"); break;
            case 2:
                if (f.wt != 0) fail("bad encoding of TensorProto.data_type");
                t.dtype = (int)std::min<uint64_t>(f.v, 1u << 20);
                break;
            case 4: repeated_f32(f, t.floats, "TensorProto.float_data"); break;
            case 5: case 7: case 11: repeated_ints(f, t.ints, "TensorProto integer data"); break;
            case 8:
                if (f.wt != 2) fail("bad encoding of TensorProto.name");
                t.name = str(f.s);
                break;
            case 9:
                if (f.wt != 2) fail("bad encoding of TensorProto.raw_data");
                t.raw = f.s;
                t.has_raw = true;
                break;
            case 10: repeated_f64(f, t.doubles, "TensorProto.double_data"); break;
            case 13: external = true; break;
            case 14:
                if (f.wt == 0 && f.v == 1) external = true;
                break;
            default: break;
        }
    }
    if (external) fail("tensor " + t.name + ": external data files are not supported (export with embedded weights)");
    const int isz = item_size(t.dtype);
    if (!isz) fail("tensor " + t.name + ": unsupported ONNX data type " + std::to_string(t.dtype));
    t.count = shape_count(t.dims, t.name);
    size_t have;
    if (t.has_raw) {
        if (t.raw.n % (size_t)isz) fail("tensor " + t.name + ": raw_data is not a whole number of elements");
        have = t.raw.n / (size_t)isz;
    } else if (t.dtype == FLOAT) have = t.floats.size();
    else if (t.dtype == DOUBLE) have = t.doubles.size();
    else have = t.ints.size();
    if (have != t.count) fail("tensor " + t.name + ": " + std::to_string(have) + " values for dims " + shape_str(t.dims));
    return t;
}

// any supported tensor -> f32 (what numpy's astype(float32) gives)
Arr to_f32(const Tensor& t) {
    Arr a;
    a.shape = t.dims;
    a.v.resize(t.count);
    float* o = a.v.data();
    const size_t n = t.count;
    if (t.has_raw) {
        const uint8_t* p = t.raw.p;
        switch (t.dtype) {
            case FLOAT: if (n) memcpy(o, p, n * 4); break;
            case UINT8: case BOOL: for (size_t i = 0; i < n; ++i) o[i] = (float)p[i]; break;
            case INT8: for (size_t i = 0; i < n; ++i) o[i] = (float)(int8_t)p[i]; break;
            case UINT16: for (size_t i = 0; i < n; ++i) o[i] = (float)load<uint16_t>(p + 2 * i); break;
            case INT16: for (size_t i = 0; i < n; ++i) o[i] = (float)load<int16_t>(p + 2 * i); break;
            case INT32: for (size_t i = 0; i < n; ++i) o[i] = (float)load<int32_t>(p + 4 * i); break;
            case UINT32: for (size_t i = 0; i < n; ++i) o[i] = (float)load<uint32_t>(p + 4 * i); break;
            case INT64: for (size_t i = 0; i < n; ++i) o[i] = (float)load<int64_t>(p + 8 * i); break;
            case UINT64: for (size_t i = 0; i < n; ++i) o[i] = (float)load<uint64_t>(p + 8 * i); break;
            case DOUBLE: for (size_t i = 0; i < n; ++i) o[i] = (float)load<double>(p + 8 * i); break;
            case FLOAT16: for (size_t i = 0; i < n; ++i) o[i] = half_to_float(load<uint16_t>(p + 2 * i)); break;
            case BFLOAT16: for (size_t i = 0; i < n; ++i) o[i] = bf16_to_float(load<uint16_t>(p + 2 * i)); break;
            default: fail("tensor " + t.name + ": unsupported ONNX data type");
        }
    } else if (t.dtype == FLOAT) {
        std::copy(t.floats.begin(), t.floats.end(), o);
    } else if (t.dtype == DOUBLE) {
        for (size_t i = 0; i < n; ++i) o[i] = (float)t.doubles[i];
    } else if (t.dtype == FLOAT16) {  // bit patterns in int32_data
        for (size_t i = 0; i < n; ++i) o[i] = half_to_float((uint16_t)t.ints[i]);
    } else if (t.dtype == BFLOAT16) {
        for (size_t i = 0; i < n; ++i) o[i] = bf16_to_float((uint16_t)t.ints[i]);
    } else {
        for (size_t i = 0; i < n; ++i) o[i] = (float)t.ints[i];
    }
    return a;
}

// the bytes of a one-byte-per-element tensor (packed 4-bit weights and zero points)
std::vector<uint8_t> to_u8(const Tensor& t) {
    if (item_size(t.dtype) != 1) fail("tensor " + t.name + ": expected 8-bit storage");
    std::vector<uint8_t> o(t.count);
    if (t.has_raw) {
        if (t.count) memcpy(o.data(), t.raw.p, t.count);
    } else {
        for (size_t i = 0; i < t.count; ++i) o[i] = (uint8_t)t.ints[i];
    }
    return o;
}

struct Attr {
    bool has_i = false, has_f = false;
    int64_t i = 0;
    float f = 0.f;
    std::vector<int64_t> ints;
    std::shared_ptr<Tensor> t;
};

struct Node {
    std::string op, name, domain;
    std::vector<std::string> in, out;
    std::map<std::string, Attr> attrs;
    int64_t attr_int(const char* k, int64_t dflt) const {
        auto it = attrs.find(k);
        if (it == attrs.end()) return dflt;
        if (it->second.has_i) return it->second.i;
        if (it->second.has_f) {
            const float v = it->second.f;
            return (v > -1e15f && v < 1e15f) ? (int64_t)v : dflt;  // (also false for NaN)
        }
        return dflt;
    }
    bool has_attr(const char* k) const { return attrs.count(k) != 0; }
};

void parse_attr(Span buf, Node& n) {
    std::string name;
    Attr a;
    Msg m(buf);
    Field f;
    while (m.next(f)) {
        switch (f.no) {
            case 1:
                if (f.wt != 2) fail("bad encoding of AttributeProto.name");
                name = str(f.s);
                break;
            case 2:
                if (f.wt != 5) fail("bad encoding of AttributeProto.f");
                a.f = load<float>(f.s.p);
                a.has_f = true;
                break;
            case 3:
                if (f.wt != 0) fail("bad encoding of AttributeProto.i");
                a.i = (int64_t)f.v;
                a.has_i = true;
                break;
            case 5:
                if (f.wt != 2) fail("bad encoding of AttributeProto.t");
                a.t = std::make_shared<Tensor>(parse_tensor(f.s));
                break;
            case 8: repeated_ints(f, a.ints, "AttributeProto.ints"); break;
            default: break;
        }
    }
    n.attrs[name] = std::move(a);
}

Node parse_node(Span buf) {
    Node n;
    Msg m(buf);
    Field f;
    while (m.next(f)) {
        if (f.no >= 1 && f.no <= 7 && f.no != 6 && f.wt != 2) fail("bad encoding of a NodeProto field");
        switch (f.no) {
            case 1: n.in.push_back(str(f.s)); break;
            case 2: n.out.push_back(str(f.s)); break;
            case 3: n.name = str(f.s); break;
            case 4: n.op = str(f.s); break;
            case 5: parse_attr(f.s, n); break;
            case 7: n.domain = str(f.s); break;
            default: break;
        }
    }
    return n;
}

struct Graph {
    std::vector<Node> nodes;
    std::vector<Tensor> inits;              // first-insertion order; a repeated name replaces the value in place
    std::map<std::string, size_t> by_name;
    void set(Tensor&& t, bool keep_existing) {
        auto it = by_name.find(t.name);
        if (it != by_name.end()) {
            if (!keep_existing) inits[it->second] = std::move(t);
            return;
        }
        by_name[t.name] = inits.size();
        inits.push_back(std::move(t));
    }
    const Tensor* find(const std::string& name) const {
        auto it = by_name.find(name);
        return it == by_name.end() ? nullptr : &inits[it->second];
    }
};

Graph read_graph(Span file) {
    Graph g;
    bool found = false;
    Msg m(file);
    Field f;
    while (m.next(f)) {
        if (f.no != 7 || f.wt != 2) continue;
        found = true;
        Msg gm(f.s);
        Field gf;
        while (gm.next(gf)) {
            if (gf.no == 1 && gf.wt == 2) g.nodes.push_back(parse_node(gf.s));
            else if (gf.no == 5 && gf.wt == 2) g.set(parse_tensor(gf.s), false);
        }
    }
    if (!found) fail("no GraphProto (not an ONNX model?)");
    // Constant nodes carry tensors too (weights folded at export time often end up there)
    for (const Node& n : g.nodes) {
        if (n.op != "Constant" || n.out.empty()) continue;
        auto it = n.attrs.find("value");
        if (it == n.attrs.end() || !it->second.t) continue;
        Tensor t = *it->second.t;
        t.name = n.out[0];
        g.set(std::move(t), true);
    }
    return g;
}

// ---- array helpers --------------------------------------------------------------------------------------------------
Arr transpose(const Arr& a, const std::vector<int64_t>& perm_in) {
    const size_t nd = a.shape.size();
    std::vector<size_t> perm(nd);
    if (perm_in.empty()) {
        for (size_t i = 0; i < nd; ++i) perm[i] = nd - 1 - i;
    } else {
        if (perm_in.size() != nd) fail("Transpose: perm does not match the tensor rank");
        std::vector<bool> seen(nd, false);
        for (size_t i = 0; i < nd; ++i) {
            if (perm_in[i] < 0 || (size_t)perm_in[i] >= nd || seen[(size_t)perm_in[i]]) fail("Transpose: bad perm");
            seen[(size_t)perm_in[i]] = true;
            perm[i] = (size_t)perm_in[i];
        }
    }
    std::vector<size_t> in_stride(nd, 1);
    for (size_t i = nd; i-- > 1;) in_stride[i - 1] = in_stride[i] * (size_t)a.shape[i];
    Arr o;
    o.shape.resize(nd);
    for (size_t i = 0; i < nd; ++i) o.shape[i] = a.shape[perm[i]];
    o.v.resize(a.v.size());
    if (a.v.empty()) return o;
    if (nd == 2 && perm[0] == 1) {  // the common case (Linear weights), cache-blocked
        const size_t R = (size_t)a.shape[0], C = (size_t)a.shape[1];
        for (size_t r0 = 0; r0 < R; r0 += 32)
            for (size_t c0 = 0; c0 < C; c0 += 32)
                for (size_t r = r0; r < std::min(R, r0 + 32); ++r)
                    for (size_t c = c0; c < std::min(C, c0 + 32); ++c) o.v[c * R + r] = a.v[r * C + c];
        return o;
    }
    std::vector<size_t> idx(nd, 0);
    for (size_t lin = 0; lin < o.v.size(); ++lin) {
        size_t src = 0;
        for (size_t i = 0; i < nd; ++i) src += idx[i] * in_stride[perm[i]];
        o.v[lin] = a.v[src];
        for (size_t i = nd; i-- > 0;) {
            if (++idx[i] < (size_t)o.shape[i]) break;
            idx[i] = 0;
        }
    }
    return o;
}

// (x - zero_point) * scale in f32: per tensor, per slice of `axis`, or blocked along `axis` (DequantizeLinear-21)
Arr dequant(const Arr& x, const Arr& scale, const Arr* zp, int64_t axis, int64_t block, const std::string& who) {
    Arr o;
    o.shape = x.shape;
    o.v.resize(x.v.size());
    const size_t nd = x.shape.size();
    if (zp && zp->v.size() != scale.v.size()) fail(who + ": zero point and scale differ in size");
    if (scale.v.size() == 1) {
        const float z = zp ? zp->v[0] : 0.f, s = scale.v[0];
        for (size_t i = 0; i < x.v.size(); ++i) o.v[i] = (x.v[i] - z) * s;
        return o;
    }
    if (axis < 0) axis += (int64_t)nd;
    if (axis < 0 || (size_t)axis >= nd) fail(who + ": quantisation axis out of range");
    size_t inner = 1;
    for (size_t i = (size_t)axis + 1; i < nd; ++i) inner *= (size_t)x.shape[i];
    const size_t A = (size_t)x.shape[(size_t)axis];
    if (block > 0) {
        if (scale.shape.size() != nd) fail(who + ": blocked scale must have the rank of the tensor");
        const size_t nb = (size_t)scale.shape[(size_t)axis];
        for (size_t i = 0; i < nd; ++i)
            if (i != (size_t)axis && scale.shape[i] != x.shape[i]) fail(who + ": blocked scale shape mismatch");
        if (nb * (size_t)block < A) fail(who + ": blocked scale too short");
        for (size_t lin = 0; lin < x.v.size(); ++lin) {
            const size_t in = lin % inner, a = (lin / inner) % A, out = lin / (inner * A);
            const size_t si = (out * nb + a / (size_t)block) * inner + in;
            o.v[lin] = (x.v[lin] - (zp ? zp->v[si] : 0.f)) * scale.v[si];
        }
        return o;
    }
    if (scale.v.size() != A) fail(who + ": per-channel scale does not match the quantisation axis");
    for (size_t lin = 0; lin < x.v.size(); ++lin) {
        const size_t a = (lin / inner) % A;
        o.v[lin] = (x.v[lin] - (zp ? zp->v[a] : 0.f)) * scale.v[a];
    }
    return o;
}

// rows in ONNX order [i o f c] -> PyTorch order [i f g o] (g = ONNX's c); `a` points at [4*hid][cols]
void lstm_gates(const float* a, size_t hid, size_t cols, float* out) {
    const size_t blk = hid * cols;
    memcpy(out, a, blk * 4);                      // i
    memcpy(out + blk, a + 2 * blk, blk * 4);      // f
    memcpy(out + 2 * blk, a + 3 * blk, blk * 4);  // g (= c)
    memcpy(out + 3 * blk, a + blk, blk * 4);      // o
}

std::string replace_all(std::string s, const std::string& from, const std::string& to) {
    for (size_t pos = 0; (pos = s.find(from, pos)) != std::string::npos; pos += to.size()) s.replace(pos, from.size(), to);
    return s;
}

// '/decoder/generator/resblocks.0/convs1.0/Conv_1' -> 'decoder.generator.resblocks.0.convs1.0'
std::string module_path(const std::string& node_name) {
    std::vector<std::string> parts;
    size_t i = 0;
    while (i <= node_name.size()) {
        size_t j = node_name.find('/', i);
        if (j == std::string::npos) j = node_name.size();
        if (j > i) parts.push_back(node_name.substr(i, j - i));
        i = j + 1;
    }
    if (parts.size() <= 1) return "";
    std::string o;
    for (size_t k = 0; k + 1 < parts.size(); ++k) o += (k ? "." : "") + parts[k];
    return o;
}

std::string list_repr(const std::vector<std::string>& v, size_t limit) {
    std::string o = "[";
    for (size_t i = 0; i < v.size() && i < limit; ++i) o += (i ? ", '" : "'") + v[i] + "'";
    return o + "]";
}

const char* const WN_SUFFIXES[4] = {".weight_g", ".weight_v", ".parametrizations.weight.original0",
                                    ".parametrizations.weight.original1"};
const char* const IGNORED_SUFFIXES[8] = {"position_ids", "pooler.weight", "pooler.bias", ".norm.weight", ".norm.bias",
                                         ".norm.running_mean", ".norm.running_var", ".norm.num_batches_tracked"};

bool ignorable(const std::string& s) {
    for (const char* suf : IGNORED_SUFFIXES)
        if (ends_with(s, suf)) return true;
    return false;
}

// insertion-ordered name -> array map (Python dict semantics)
struct Flat {
    std::vector<std::pair<std::string, Arr>> items;
    std::map<std::string, size_t> idx;
    bool has(const std::string& k) const { return idx.count(k) != 0; }
    Arr* get(const std::string& k) {
        auto it = idx.find(k);
        return it == idx.end() ? nullptr : &items[it->second].second;
    }
    void set(const std::string& k, Arr&& a) {
        auto it = idx.find(k);
        if (it != idx.end()) {
            items[it->second].second = std::move(a);
            return;
        }
        idx[k] = items.size();
        items.emplace_back(k, std::move(a));
    }
};

struct Importer {
    const Graph& g;
    const std::vector<SpecEntry>& spec;
    std::map<std::string, const SpecEntry*> spec_by_name;
    std::map<std::string, Arr> derived;  // tensors computed from initialisers by Cast / Transpose / DequantizeLinear
    std::set<std::string> used;
    Flat flat;

    explicit Importer(const Graph& gr) : g(gr), spec(tensor_spec()) {
        for (const SpecEntry& e : spec) spec_by_name[e.name] = &e;
    }

    bool is_const(const std::string& name) const { return derived.count(name) || g.find(name); }

    // f32 view of a constant (derived tensor or initialiser); false when `name` is neither
    bool konst(const std::string& name, Arr& out) {
        auto d = derived.find(name);
        if (d != derived.end()) {
            out = d->second;
            return true;
        }
        const Tensor* t = g.find(name);
        if (!t) return false;
        used.insert(name);
        out = to_f32(*t);
        return true;
    }
    // rank of a constant without converting it (an initialiser that is looked at counts as used, like konst)
    int konst_rank(const std::string& name) {
        auto d = derived.find(name);
        if (d != derived.end()) return (int)d->second.shape.size();
        const Tensor* t = g.find(name);
        if (t) used.insert(name);
        return t ? (int)t->dims.size() : -1;
    }

    bool in_spec(const std::string& n) const { return spec_by_name.count(n) != 0; }

    bool unique_tail(const std::string& name, std::string& out) const {
        const std::string suf = "." + name;
        int hits = 0;
        for (const SpecEntry& e : spec)
            if (ends_with(e.name, suf)) {
                if (++hits > 1) return false;
                out = e.name;
            }
        return hits == 1;
    }

    // the table (or weight-norm / ignorable) name an initialiser name stands for, tolerating one extra or one missing
    // leading component (exporters wrap the model: 'kmodel.bert...' / 'encoder...')
    bool canonical(const std::string& name, std::string& out) const {
        std::vector<std::string> cands{name};
        const size_t dot = name.find('.');
        if (dot != std::string::npos) cands.push_back(name.substr(dot + 1));
        for (const std::string& c : cands) {
            std::string base = c;
            for (const char* suf : WN_SUFFIXES)
                if (ends_with(c, suf)) base = c.substr(0, c.size() - strlen(suf)) + ".weight";
            if (in_spec(base) || ignorable(c)) {
                out = c;
                return true;
            }
        }
        return unique_tail(name, out);
    }

    // module path from a node name -> full table name, also when the path lacks leading components or carries up to two
    // extra ones (exporters wrap the model: '/kmodel/bert/...', '/model/kmodel/bert/...')
    bool spec_name(const std::string& path, const std::string& leaf, std::string& out) const {
        const std::string full = path.empty() ? leaf : path + "." + leaf;
        std::string cand = full;
        for (int strip = 0; strip < 3; ++strip) {
            if (in_spec(cand)) {
                out = cand;
                return true;
            }
            const size_t dot = cand.find('.');
            if (dot == std::string::npos) break;
            cand = cand.substr(dot + 1);
        }
        return unique_tail(full, out);
    }

    void place(const std::string& name, Arr&& a) {
        auto sp = spec_by_name.find(name);
        if (sp != spec_by_name.end()) {
            const SpecEntry& e = *sp->second;
            Shape want(e.dims, e.dims + e.ndim);
            if (a.shape.size() == want.size() + 1 && std::find(a.shape.begin(), a.shape.end(), 1) != a.shape.end()) {
                Shape s;  // 1-D convs exported as 2-D ones: drop a unit dim at position 2
                for (size_t i = 0; i < a.shape.size(); ++i)
                    if (!(a.shape[i] == 1 && i == 2)) s.push_back(a.shape[i]);
                a.shape = s;
            }
            if (a.shape != want && a.size() == e.count()) a.shape = want;  // e.g. alpha [ch] vs [1, ch, 1]
        }
        if (!flat.has(name)) flat.set(name, std::move(a));
    }

    Arr nbits_weight(const Node& n) {
        const int64_t K = n.attr_int("K", -1), N = n.attr_int("N", -1), bits = n.attr_int("bits", 4),
                      bs = n.attr_int("block_size", -1);
        if (bits != 4) fail(n.name + ": MatMulNBits with bits=" + std::to_string(bits) + " is not supported");
        if (K <= 0 || N <= 0 || bs < 2 || (bs & 1) || K > (1 << 24) || N > (1 << 24) || bs > (1 << 20))
            fail(n.name + ": MatMulNBits needs positive K, N and an even block_size");
        if (n.in.size() < 3) fail(n.name + ": MatMulNBits needs B and scales");
        const Tensor* tb = g.find(n.in[1]);
        const Tensor* ts = g.find(n.in[2]);
        if (!tb || !ts) fail(n.name + ": MatMulNBits weights are not initialisers");
        used.insert(n.in[1]);
        used.insert(n.in[2]);
        const size_t nb = (size_t)((K + bs - 1) / bs), half = (size_t)bs / 2, Nn = (size_t)N;
        const std::vector<uint8_t> B = to_u8(*tb);
        if (B.size() != Nn * nb * half) fail(n.name + ": packed weight size does not match N, K, block_size");
        const Arr sc = to_f32(*ts);
        if (sc.size() != Nn * nb) fail(n.name + ": scales size does not match N, K, block_size");
        std::vector<uint8_t> zraw;
        size_t zcols = 0;
        if (n.in.size() > 3 && !n.in[3].empty()) {
            const Tensor* tz = g.find(n.in[3]);
            if (!tz) fail(n.name + ": MatMulNBits zero points are not an initialiser");
            used.insert(n.in[3]);
            zraw = to_u8(*tz);
            if (zraw.size() % Nn) fail(n.name + ": zero points do not divide into N rows");
            zcols = zraw.size() / Nn;
            if (zcols * 2 < nb) fail(n.name + ": too few zero points");
        }
        Arr w;
        w.shape = {N, K};
        w.v.resize(Nn * (size_t)K);
        for (size_t r = 0; r < Nn; ++r)
            for (size_t b = 0; b < nb; ++b) {
                float zp = 8.0f;
                if (!zraw.empty()) {
                    const uint8_t z = zraw[r * zcols + b / 2];
                    zp = (float)((b & 1) ? (z >> 4) : (z & 0x0F));
                }
                const float s = sc.v[r * nb + b];
                const uint8_t* src = &B[(r * nb + b) * half];
                for (size_t j = 0; j < (size_t)bs; ++j) {
                    const size_t k = b * (size_t)bs + j;
                    if (k >= (size_t)K) break;
                    const float q = (float)((j & 1) ? (src[j / 2] >> 4) : (src[j / 2] & 0x0F));
                    w.v[r * (size_t)K + k] = (q - zp) * s;
                }
            }
        return w;
    }

    void derive_constants() {
        for (const Node& n : g.nodes) {
            if (n.out.empty() || n.in.empty()) continue;
            if (n.op == "DequantizeLinear" && g.find(n.in[0])) {
                Arr x, sc, zp;
                if (n.in.size() < 2 || !konst(n.in[1], sc)) fail(n.name + ": DequantizeLinear scale is not an initialiser");
                konst(n.in[0], x);
                const bool has_zp = n.in.size() > 2 && !n.in[2].empty() && konst(n.in[2], zp);
                derived[n.out[0]] = dequant(x, sc, has_zp ? &zp : nullptr, n.attr_int("axis", 1), n.attr_int("block_size", 0),
                                            n.name.empty() ? std::string("DequantizeLinear") : n.name);
            } else if (n.op == "Cast" && is_const(n.in[0])) {
                Arr a;
                konst(n.in[0], a);
                derived[n.out[0]] = std::move(a);
            } else if (n.op == "Transpose" && is_const(n.in[0])) {
                Arr a;
                konst(n.in[0], a);
                auto it = n.attrs.find("perm");
                derived[n.out[0]] = transpose(a, it == n.attrs.end() ? std::vector<int64_t>{} : it->second.ints);
            }
        }
    }

    void place_named() {
        for (const Tensor& t : g.inits) {
            std::string c;
            if (!canonical(t.name, c)) continue;
            if (t.dtype != FLOAT && t.dtype != FLOAT16 && t.dtype != BFLOAT16 && t.dtype != DOUBLE) continue;
            Arr a;
            konst(t.name, a);
            place(c, std::move(a));
        }
    }

    void place_through_nodes() {
        std::map<std::string, std::vector<const Node*>> consumers;
        for (const Node& n : g.nodes)
            for (const std::string& i : n.in) consumers[i].push_back(&n);
        std::string tmp;
        for (const Node& n : g.nodes) {
            const std::string path = module_path(n.name);
            const std::string& op = n.op;
            if (op == "Conv" || op == "ConvTranspose" || op == "ConvInteger") {
                if (n.in.size() < 2) continue;
                const std::string& wname = n.in[1];
                if (!canonical(wname, tmp)) {
                    Arr a;
                    bool have = konst(wname, a);
                    if (op == "ConvInteger" && have) {
                        Arr zp, sc;
                        const bool has_zp = n.in.size() > 3 && !n.in[3].empty() && konst(n.in[3], zp);
                        const std::string sname = replace_all(wname, "_quantized", "_scale");
                        if (!konst(sname, sc)) fail(n.name + ": no '" + sname + "' initialiser for ConvInteger");
                        a = dequant(a, sc, has_zp ? &zp : nullptr, 0, 0, n.name);
                    }
                    std::string tgt;
                    if (have && spec_name(path, "weight", tgt)) place(tgt, std::move(a));
                }
                // (ConvInteger's third input is the activation zero point, not a bias)
                if (op != "ConvInteger" && n.in.size() > 2 && !n.in[2].empty() && !canonical(n.in[2], tmp)) {
                    Arr b;
                    std::string tgt;
                    if (konst(n.in[2], b) && spec_name(path, "bias", tgt)) place(tgt, std::move(b));
                }
            } else if (op == "MatMul" || op == "MatMulInteger" || op == "MatMulNBits" || op == "Gemm") {
                if (n.in.size() < 2 || n.out.empty()) continue;
                Arr wt;
                bool have = false;
                if (op == "MatMulNBits") {
                    wt = nbits_weight(n);
                    have = true;
                } else if (op == "MatMulInteger") {
                    Arr bq;
                    if (konst(n.in[1], bq)) {
                        Arr zp, sc;
                        const bool has_zp = n.in.size() > 3 && !n.in[3].empty() && konst(n.in[3], zp);
                        if (!konst(replace_all(n.in[1], "_quantized", "_scale"), sc))
                            fail(n.name + ": no scale initialiser beside " + n.in[1]);
                        if (bq.shape.size() != 2) fail(n.name + ": MatMulInteger weight is not a matrix");
                        wt = transpose(dequant(bq, sc, has_zp ? &zp : nullptr, 1, 0, n.name), {});
                        have = true;
                    }
                } else {
                    Arr b;
                    if (konst_rank(n.in[1]) == 2 && konst(n.in[1], b)) {
                        wt = (op == "Gemm" && n.attr_int("transB", 0)) ? std::move(b) : transpose(b, {});
                        have = true;
                    }
                }
                if (!have) continue;
                std::string tgt;
                bool have_tgt = spec_name(path, "weight", tgt);
                Arr bias;
                std::string bias_name;
                bool have_bias = false;
                if (op == "Gemm" && n.in.size() > 2 && !n.in[2].empty()) {
                    have_bias = konst(n.in[2], bias);
                    bias_name = n.in[2];
                } else {  // the Add that follows a MatMul carries the (named) bias
                    std::vector<std::string> outs{n.out[0]};
                    for (int hop = 0; hop < 3; ++hop) {  // MatMulInteger: Cast / Mul sit between the product and the Add
                        std::vector<const Node*> nxt;
                        for (const std::string& o : outs) {
                            auto it = consumers.find(o);
                            if (it != consumers.end()) nxt.insert(nxt.end(), it->second.begin(), it->second.end());
                        }
                        const Node* add = nullptr;
                        for (const Node* c : nxt)
                            if (c->op == "Add") {
                                add = c;
                                break;
                            }
                        if (add) {
                            for (const std::string& i : add->in) {
                                if (std::find(outs.begin(), outs.end(), i) != outs.end()) continue;
                                if (konst_rank(i) == 1 && konst(i, bias)) {
                                    have_bias = true;
                                    bias_name = i;
                                }
                                break;  // only the first other input is looked at
                            }
                            break;
                        }
                        std::vector<std::string> next_outs;
                        for (const Node* c : nxt)
                            if ((c->op == "Cast" || c->op == "Mul") && !c->out.empty()) next_outs.push_back(c->out[0]);
                        outs = next_outs;
                        if (outs.empty()) break;
                    }
                }
                std::string cb;
                if (!have_tgt && have_bias && canonical(bias_name, cb) && ends_with(cb, ".bias")) {
                    tgt = cb.substr(0, cb.size() - 5) + ".weight";
                    have_tgt = true;
                }
                if (have_tgt) {
                    auto sp = spec_by_name.find(tgt);
                    if (sp != spec_by_name.end() && wt.size() == sp->second->count()) {
                        place(tgt, std::move(wt));
                        if (have_bias && !canonical(bias_name, cb)) place(tgt.substr(0, tgt.size() - 7) + ".bias", std::move(bias));
                    }
                }
            } else if (op == "LSTM") {
                if (n.in.size() < 3) continue;
                Arr Wi, R, B;
                if (!konst(n.in[1], Wi) || !konst(n.in[2], R)) continue;
                const bool has_b = n.in.size() > 3 && !n.in[3].empty() && konst(n.in[3], B);
                if (Wi.shape.size() != 3 || R.shape.size() != 3) fail(n.name + ": LSTM W / R are not [directions][4*hidden][n]");
                const int64_t hid = n.attr_int("hidden_size", Wi.shape[1] / 4);
                std::string base;
                if (!spec_name(path, "weight_ih_l0", base)) continue;
                base = base.substr(0, base.size() - strlen(".weight_ih_l0"));
                const int64_t dirs = Wi.shape[0], n_in = Wi.shape[2];
                if (hid <= 0 || hid > (1 << 24) || Wi.shape[1] != 4 * hid || R.shape[0] != dirs || R.shape[1] != 4 * hid || R.shape[2] != hid)
                    fail(n.name + ": LSTM W / R shapes do not match hidden_size");
                if (has_b && (B.shape.size() != 2 || B.shape[0] != dirs || B.shape[1] != 8 * hid))
                    fail(n.name + ": LSTM B is not [directions][8*hidden]");
                const size_t H = (size_t)hid;
                for (int64_t d = 0; d < dirs; ++d) {
                    const std::string suf = d == 0 ? "" : "_reverse";
                    Arr a;
                    a.shape = {4 * hid, n_in};
                    a.v.resize(4 * H * (size_t)n_in);
                    lstm_gates(Wi.v.data() + (size_t)d * 4 * H * (size_t)n_in, H, (size_t)n_in, a.v.data());
                    place(base + ".weight_ih_l0" + suf, std::move(a));
                    Arr r;
                    r.shape = {4 * hid, hid};
                    r.v.resize(4 * H * H);
                    lstm_gates(R.v.data() + (size_t)d * 4 * H * H, H, H, r.v.data());
                    place(base + ".weight_hh_l0" + suf, std::move(r));
                    if (has_b) {
                        for (int half = 0; half < 2; ++half) {
                            Arr b;
                            b.shape = {4 * hid};
                            b.v.resize(4 * H);
                            lstm_gates(B.v.data() + (size_t)d * 8 * H + (size_t)half * 4 * H, H, 1, b.v.data());
                            place(base + (half ? ".bias_hh_l0" : ".bias_ih_l0") + suf, std::move(b));
                        }
                    }
                }
            }
        }
    }

    // x.weight = g * v / ||v|| (norm over every dim but 0, in f64, summed in index order), for weight_g / weight_v and
    // the parametrizations spelling; a DataParallel "module." prefix is dropped first
    Flat fold_weight_norm() {
        Flat stripped;
        for (auto& kv : flat.items)
            stripped.set(starts_with(kv.first, "module.") ? kv.first.substr(7) : kv.first, std::move(kv.second));
        Flat out;
        struct Pair { Arr *g = nullptr, *v = nullptr; };
        std::vector<std::pair<std::string, Pair>> pairs;
        std::map<std::string, size_t> pidx;
        auto pair_of = [&](const std::string& base) -> Pair& {
            auto it = pidx.find(base);
            if (it == pidx.end()) {
                pidx[base] = pairs.size();
                pairs.emplace_back(base, Pair{});
                return pairs.back().second;
            }
            return pairs[it->second].second;
        };
        for (auto& kv : stripped.items) {
            const std::string& k = kv.first;
            bool taken = false;
            for (int s = 0; s < 4 && !taken; s += 2) {
                if (ends_with(k, WN_SUFFIXES[s])) {
                    pair_of(k.substr(0, k.size() - strlen(WN_SUFFIXES[s]))).g = &kv.second;
                    taken = true;
                } else if (ends_with(k, WN_SUFFIXES[s + 1])) {
                    pair_of(k.substr(0, k.size() - strlen(WN_SUFFIXES[s + 1]))).v = &kv.second;
                    taken = true;
                }
            }
            if (!taken) out.set(k, std::move(kv.second));
        }
        for (auto& bp : pairs) {
            const Pair& p = bp.second;
            if (!p.g || !p.v) fail("incomplete weight-norm pair for " + bp.first);
            const Arr& v = *p.v;
            const Arr& gq = *p.g;
            if (v.shape.empty() || v.shape[0] <= 0) fail("weight-norm pair for " + bp.first + ": empty direction tensor");
            const size_t rows = (size_t)v.shape[0], cols = v.size() / rows;
            if (gq.size() != rows && gq.size() != 1) fail("weight-norm pair for " + bp.first + ": magnitude does not match the rows");
            Arr w;
            w.shape = v.shape;
            w.v.resize(v.size());
            for (size_t r = 0; r < rows; ++r) {
                double ss = 0.0;
                for (size_t c = 0; c < cols; ++c) {
                    const double x = (double)v.v[r * cols + c];
                    ss += x * x;
                }
                const double norm = std::sqrt(ss), gg = (double)gq.v[gq.size() == 1 ? 0 : r];
                for (size_t c = 0; c < cols; ++c) w.v[r * cols + c] = (float)(gg * (double)v.v[r * cols + c] / norm);
            }
            out.set(bp.first + ".weight", std::move(w));
        }
        return out;
    }

    std::vector<std::string> unused_report() const {
        std::vector<std::string> u;
        for (const Tensor& t : g.inits)
            if (!used.count(t.name) && t.count > 16) u.push_back(t.name);
        std::sort(u.begin(), u.end());
        if (u.size() > 50) u.resize(50);
        return u;
    }
};

void put_u32(std::vector<unsigned char>& b, size_t at, uint32_t v) { memcpy(&b[at], &v, 4); }
void put_u64(std::vector<unsigned char>& b, size_t at, uint64_t v) { memcpy(&b[at], &v, 8); }

}  // namespace

const std::vector<SpecEntry>& tensor_spec() {
    static const std::vector<SpecEntry> spec = [] {
        // hyper-parameters: SURVEY.md Appendix A.1 (same constants as kokorox_amd/weights.py)
        const int N_TOKEN = 178, HIDDEN = 512, STYLE = 128, MAX_DUR = 50, BERT_H = 768, BERT_E = 128, BERT_FF = 2048,
                  BERT_MAXPOS = 512, N_FFT = 20, HARMONICS = 9;
        std::vector<SpecEntry> s;
        auto add = [&](const std::string& name, std::initializer_list<int> dims) {
            SpecEntry e;
            e.name = name;
            e.ndim = (int)dims.size();
            int i = 0;
            for (int k = 0; k < 4; ++k) e.dims[k] = 0;
            for (int d : dims) e.dims[i++] = d;
            s.push_back(e);
        };
        auto lstm = [&](const std::string& p, int n_in) {
            const int hid = HIDDEN / 2;
            for (const char* suf : {"", "_reverse"}) {
                add(p + ".weight_ih_l0" + suf, {4 * hid, n_in});
                add(p + ".weight_hh_l0" + suf, {4 * hid, hid});
                add(p + ".bias_ih_l0" + suf, {4 * hid});
                add(p + ".bias_hh_l0" + suf, {4 * hid});
            }
        };
        auto adain_fc = [&](const std::string& p, int ch) {
            add(p + ".fc.weight", {2 * ch, STYLE});
            add(p + ".fc.bias", {2 * ch});
        };
        auto conv = [&](const std::string& p, int cout, int cin, int k, bool bias = true) {
            add(p + ".weight", {cout, cin, k});
            if (bias) add(p + ".bias", {cout});
        };
        auto adain_resblk = [&](const std::string& p, int din, int dout, bool upsample = false) {
            conv(p + ".conv1", dout, din, 3);
            conv(p + ".conv2", dout, dout, 3);
            adain_fc(p + ".norm1", din);
            adain_fc(p + ".norm2", dout);
            if (din != dout) conv(p + ".conv1x1", dout, din, 1, false);
            if (upsample) {  // depth-wise ConvTranspose1d(din, din, k3, s2, groups=din)
                add(p + ".pool.weight", {din, 1, 3});
                add(p + ".pool.bias", {din});
            }
        };
        auto adain_resblock1 = [&](const std::string& p, int ch, int k) {
            for (int i = 0; i < 3; ++i) {
                const std::string n = std::to_string(i);
                conv(p + ".convs1." + n, ch, ch, k);
                conv(p + ".convs2." + n, ch, ch, k);
                adain_fc(p + ".adain1." + n, ch);
                adain_fc(p + ".adain2." + n, ch);
                add(p + ".alpha1." + n, {1, ch, 1});
                add(p + ".alpha2." + n, {1, ch, 1});
            }
        };
        // PL-BERT (ALBERT, one shared layer)
        const std::string e = "bert.embeddings";
        add(e + ".word_embeddings.weight", {N_TOKEN, BERT_E});
        add(e + ".position_embeddings.weight", {BERT_MAXPOS, BERT_E});
        add(e + ".token_type_embeddings.weight", {2, BERT_E});
        add(e + ".LayerNorm.weight", {BERT_E});
        add(e + ".LayerNorm.bias", {BERT_E});
        add("bert.encoder.embedding_hidden_mapping_in.weight", {BERT_H, BERT_E});
        add("bert.encoder.embedding_hidden_mapping_in.bias", {BERT_H});
        const std::string l = "bert.encoder.albert_layer_groups.0.albert_layers.0";
        for (const char* nm : {"query", "key", "value", "dense"}) {
            add(l + ".attention." + nm + ".weight", {BERT_H, BERT_H});
            add(l + ".attention." + nm + ".bias", {BERT_H});
        }
        add(l + ".attention.LayerNorm.weight", {BERT_H});
        add(l + ".attention.LayerNorm.bias", {BERT_H});
        add(l + ".ffn.weight", {BERT_FF, BERT_H});
        add(l + ".ffn.bias", {BERT_FF});
        add(l + ".ffn_output.weight", {BERT_H, BERT_FF});
        add(l + ".ffn_output.bias", {BERT_H});
        add(l + ".full_layer_layer_norm.weight", {BERT_H});
        add(l + ".full_layer_layer_norm.bias", {BERT_H});
        add("bert_encoder.weight", {HIDDEN, BERT_H});
        add("bert_encoder.bias", {HIDDEN});
        // ProsodyPredictor
        for (int i = 0; i < 3; ++i) {
            lstm("predictor.text_encoder.lstms." + std::to_string(2 * i), HIDDEN + STYLE);
            add("predictor.text_encoder.lstms." + std::to_string(2 * i + 1) + ".fc.weight", {2 * HIDDEN, STYLE});
            add("predictor.text_encoder.lstms." + std::to_string(2 * i + 1) + ".fc.bias", {2 * HIDDEN});
        }
        lstm("predictor.lstm", HIDDEN + STYLE);
        add("predictor.duration_proj.linear_layer.weight", {MAX_DUR, HIDDEN});
        add("predictor.duration_proj.linear_layer.bias", {MAX_DUR});
        lstm("predictor.shared", HIDDEN + STYLE);
        for (const char* br : {"F0", "N"}) {
            const std::string p = std::string("predictor.") + br;
            adain_resblk(p + ".0", HIDDEN, HIDDEN);
            adain_resblk(p + ".1", HIDDEN, HIDDEN / 2, true);
            adain_resblk(p + ".2", HIDDEN / 2, HIDDEN / 2);
        }
        add("predictor.F0_proj.weight", {1, HIDDEN / 2, 1});
        add("predictor.F0_proj.bias", {1});
        add("predictor.N_proj.weight", {1, HIDDEN / 2, 1});
        add("predictor.N_proj.bias", {1});
        // TextEncoder
        add("text_encoder.embedding.weight", {N_TOKEN, HIDDEN});
        for (int i = 0; i < 3; ++i) {
            const std::string p = "text_encoder.cnn." + std::to_string(i);
            conv(p + ".0", HIDDEN, HIDDEN, 5);
            add(p + ".1.gamma", {HIDDEN});
            add(p + ".1.beta", {HIDDEN});
        }
        lstm("text_encoder.lstm", HIDDEN);
        // Decoder
        adain_resblk("decoder.encode", HIDDEN + 2, 1024);
        for (int i = 0; i < 3; ++i) adain_resblk("decoder.decode." + std::to_string(i), 1024 + 2 + 64, 1024);
        adain_resblk("decoder.decode.3", 1024 + 2 + 64, 512, true);
        add("decoder.F0_conv.weight", {1, 1, 3});
        add("decoder.F0_conv.bias", {1});
        add("decoder.N_conv.weight", {1, 1, 3});
        add("decoder.N_conv.bias", {1});
        conv("decoder.asr_res.0", 64, HIDDEN, 1);
        const std::string g = "decoder.generator";
        add(g + ".m_source.l_linear.weight", {1, HARMONICS});
        add(g + ".m_source.l_linear.bias", {1});
        conv(g + ".noise_convs.0", 256, N_FFT + 2, 12);
        conv(g + ".noise_convs.1", 128, N_FFT + 2, 1);
        adain_resblock1(g + ".noise_res.0", 256, 7);
        adain_resblock1(g + ".noise_res.1", 128, 11);
        add(g + ".ups.0.weight", {512, 256, 20});  // ConvTranspose1d layout [Cin][Cout][k]
        add(g + ".ups.0.bias", {256});
        add(g + ".ups.1.weight", {256, 128, 12});
        add(g + ".ups.1.bias", {128});
        const int res_k[3] = {3, 7, 11}, res_ch[2] = {256, 128};
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 3; ++j) adain_resblock1(g + ".resblocks." + std::to_string(i * 3 + j), res_ch[i], res_k[j]);
        add(g + ".conv_post.weight", {N_FFT + 2, 128, 7});
        add(g + ".conv_post.bias", {N_FFT + 2});
        return s;
    }();
    return spec;
}

std::vector<unsigned char> onnx_to_kxw(const unsigned char* data, size_t n, ImportInfo* info) {
    const Graph g = read_graph(Span{data, n});
    if (info) {
        *info = ImportInfo{};
        for (const Tensor& t : g.inits) {
            if (t.dtype == FLOAT || t.dtype == DOUBLE) info->n_float += 1;
            else if (t.dtype == FLOAT16) info->n_half += 1;
            else if (t.dtype == BFLOAT16) info->n_bfloat += 1;
            else if (t.dtype == INT8 || t.dtype == UINT8) info->n_int8 += 1;
        }
        for (const Node& nd : g.nodes) {
            if (nd.op == "MatMulInteger" || nd.op == "ConvInteger" || nd.op == "DynamicQuantizeLinear") info->n_integer_ops += 1;
            else if (nd.op == "MatMulNBits") info->n_nbits_ops += 1;
            else if (nd.op == "DequantizeLinear") info->n_dequant_ops += 1;
        }
        // (MatMulNBits stores its 4-bit blocks as uint8 initialisers: they are not "8-bit weights")
        if (info->n_nbits_ops > 0) {
            info->n_int4_blocks = info->n_nbits_ops;
            info->n_int8 = info->n_int8 > 2 * info->n_nbits_ops ? info->n_int8 - 2 * info->n_nbits_ops : 0;
        }
    }
    Importer im(g);
    im.derive_constants();
    im.place_named();
    im.place_through_nodes();
    Flat folded = im.fold_weight_norm();
    const std::vector<SpecEntry>& spec = im.spec;
    std::vector<std::string> missing;
    for (const SpecEntry& e : spec) {
        Arr* a = folded.get(e.name);
        if (!a) {
            missing.push_back(e.name);
            continue;
        }
        if (a->shape != Shape(e.dims, e.dims + e.ndim))
            fail(e.name + ": checkpoint shape " + shape_str(a->shape) + " != expected " + shape_str(Shape(e.dims, e.dims + e.ndim)));
    }
    if (!missing.empty())
        fail(std::to_string(missing.size()) + " tensors missing from the checkpoint, e.g. " + list_repr(missing, 5) +
             "; initialisers the importer could not place: " + list_repr(im.unused_report(), 10));
    // container layout: kokorox_amd/weights.py (64-byte header, 128-byte table entries, 256-byte aligned tensors)
    const size_t nt = spec.size();
    const size_t data_off = (64 + nt * 128 + 255) / 256 * 256;
    size_t off = data_off;
    std::vector<size_t> offs(nt);
    for (size_t i = 0; i < nt; ++i) {
        offs[i] = off;
        off = (off + spec[i].count() * 4 + 255) / 256 * 256;
    }
    std::vector<unsigned char> blob(off, 0);
    memcpy(blob.data(), "KXHIPW01", 8);
    put_u32(blob, 8, (uint32_t)nt);
    put_u32(blob, 12, (uint32_t)(nt * 128));
    put_u64(blob, 16, data_off);
    put_u64(blob, 24, off);
    for (size_t i = 0; i < nt; ++i) {
        const SpecEntry& e = spec[i];
        const size_t at = 64 + i * 128;
        if (e.name.size() >= 88) fail("internal: tensor name too long: " + e.name);
        memcpy(&blob[at], e.name.data(), e.name.size());
        put_u32(blob, at + 88, 0);
        put_u32(blob, at + 92, (uint32_t)e.ndim);
        for (int k = 0; k < 4; ++k) put_u32(blob, at + 96 + 4 * (size_t)k, (uint32_t)e.dims[k]);
        put_u64(blob, at + 112, offs[i]);
        put_u64(blob, at + 120, e.count() * 4);
        const Arr* a = folded.get(e.name);
        if (e.count()) memcpy(&blob[offs[i]], a->v.data(), e.count() * 4);
    }
    return blob;
}

}  // namespace kx
