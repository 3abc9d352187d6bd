// sam_pickle.cpp -- streaming import of a static automaton pickled by the REFERENCE's dump_sam (SO/sam/utils.py:20-22 = pickle.dump of a
// StaticSAM object: `states` = a list of SAMState dataclass instances {next: dict, link, length, cnt_endpos | min_endpos}, plus
// `states_topk_next`, `input_ids` (S variant) and a handful of scalars; class paths samd_sam_only.sam.static_sam.StaticSAM[.SAMState] /
// samd.sam.static_sam.StaticSAM[.SAMState]).  The published automata (sam_alpaca_vicuna-7b-v1.3.pkl, SO/sam/utils.py:24-39) hold 20-35 M
// states; CPython's unpickler materialises three objects per state and keeps every one of them alive in its memo until the load ends
// (tens of GB, minutes), before the native builder sees a single number.  This reader executes the pickle's opcode stream itself --
// the subset protocols 2-5 emit for that object graph -- and SINKS the graph as it goes: at a SAMState's BUILD its state dict and its
// `next` dict are appended to flat int32 tables and freed; the `states` list only counts; `states_topk_next` (re-derived by the builder:
// SO/sam/static_sam.py:137-146 is a pure function of the states) is discarded element by element; `input_ids` goes straight into the
// text array.  Peak memory = the tables (~28 B per state + 8 B per edge) + the 64-byte node image layout() makes of them.  Nothing in
// the stream is executed BY THIS READER: a GLOBAL is a name, REDUCE / NEWOBJ build inert records.  Anything outside the subset returns
// SAMD_E_IO with the opcode named.  (The Python binding's fallback for such a file is pickle.load, which DOES execute the stream: the
// "runs no code" property belongs to this reader, not to samd_sam_only.sam.utils.load_reference_pickle as a whole.)
// Ownership: containers own their items and the graph dump_sam writes is a tree, so a box has ONE reference -- the stack slot or the
// container that holds it.  The memo is the only way to make a second one, so BINGET / LONG_BINGET of a container or object is refused
// (a real dump_sam stream only GETs strings and class names); every dereference of a box still checks that it is live and long enough
// (round 6: a crafted stream could free a memoized list through the discarded `states_topk_next` sink and BUILD on the dead box).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>
#include "samd_common.h"

namespace {

enum Tag : uint8_t { T_NONE, T_BOOL, T_INT, T_FLOAT, T_STR, T_MARK, T_TUPLE, T_LIST, T_DICT, T_GLOBAL, T_OBJ, T_STATE, T_DEAD, T_OPAQUE };
struct Val {
    Tag tag; union { int64_t i; double f; };
    Val() : tag(T_NONE), i(0) {}
    Val(Tag t, int64_t v) : tag(t), i(v) {}
};
enum Role : uint8_t { R_NORMAL, R_SINK_STATES, R_SINK_DISCARD, R_SINK_TEXT };
struct Box { std::vector<Val> items; int64_t memo_id = -1; int64_t count = 0; Role role = R_NORMAL; bool live = false; };

struct Reader {
    FILE *f = nullptr;
    std::vector<unsigned char> buf; size_t pos = 0, end = 0; bool eof = false;
    explicit Reader(FILE *fp) : f(fp), buf(1 << 20) {}
    bool fill() { if (eof) return false; end = fread(buf.data(), 1, buf.size(), f); pos = 0; if (end == 0) { eof = true; return false; } return true; }
    bool byte(unsigned char &b) { if (pos == end && !fill()) return false; b = buf[pos++]; return true; }
    bool bytes(void *dst, size_t n) {
        unsigned char *d = (unsigned char *)dst;
        while (n) { if (pos == end && !fill()) return false; size_t k = end - pos < n ? end - pos : n; memcpy(d, buf.data() + pos, k); pos += k; d += k; n -= k; }
        return true;
    }
    bool skip(uint64_t n) { while (n) { if (pos == end && !fill()) return false; size_t k = end - pos < n ? end - pos : (size_t)n; pos += k; n -= k; } return true; }
    bool line(std::string &s) { s.clear(); unsigned char b; while (byte(b)) { if (b == '\n') return true; s.push_back((char)b); if (s.size() > 4096) return false; } return false; }
};

struct VM {
    std::vector<Val> stack;
    std::vector<Box> boxes; std::vector<int64_t> free_boxes;
    std::vector<std::string> strs;
    std::unordered_map<int64_t, Val> memo; int64_t memo_next = 0;
    // sinks
    std::vector<int32_t> link, length, aux, deg, etok, edst, text;
    int64_t n_states = 0; bool have_text = false; int aux_is_minend = -1;
    std::string err;
    int64_t peak_boxes = 0;

    bool fail(const std::string &m) { if (err.empty()) err = m; return false; }
    int64_t new_box(Role r = R_NORMAL) {
        int64_t id;
        if (!free_boxes.empty()) { id = free_boxes.back(); free_boxes.pop_back(); }
        else { id = (int64_t)boxes.size(); boxes.emplace_back(); if ((int64_t)boxes.size() > peak_boxes) peak_boxes = (int64_t)boxes.size(); }
        Box &b = boxes[id]; b.items.clear(); b.memo_id = -1; b.count = 0; b.role = r; b.live = true;
        return id;
    }
    static bool is_box(const Val &v) { return v.tag == T_TUPLE || v.tag == T_LIST || v.tag == T_DICT || v.tag == T_OBJ; }
    // the live box behind v with at least min_items items, or nullptr (never trust an index that came through the stream's own bookkeeping)
    Box *box_of(const Val &v, size_t min_items = 0) {
        if (!is_box(v) || v.i < 0 || v.i >= (int64_t)boxes.size()) return nullptr;
        Box &b = boxes[v.i];
        return b.live && b.items.size() >= min_items ? &b : nullptr;
    }
    void free_val(const Val &v) {                           // recursive: containers own their items (the graph the reference dumps is a tree)
        if (!is_box(v)) return;
        std::vector<int64_t> todo{v.i};
        while (!todo.empty()) {
            const int64_t id = todo.back(); todo.pop_back();
            if (id < 0 || id >= (int64_t)boxes.size() || !boxes[id].live) continue;
            Box &b = boxes[id];
            for (const Val &x : b.items) if (is_box(x)) todo.push_back(x.i);
            if (b.memo_id >= 0) memo.erase(b.memo_id);      // a later GET of it is an error (never happens for this graph)
            b.live = false;
            std::vector<Val>().swap(b.items);
            free_boxes.push_back(id);
        }
    }
    int64_t str_id(std::string &&s) { strs.push_back(std::move(s)); return (int64_t)strs.size() - 1; }
    bool str_is(const Val &v, const char *s) const { return v.tag == T_STR && strs[v.i] == s; }
    bool global_ends_with(const Val &v, const char *suffix) const {
        if (v.tag != T_GLOBAL) return false;
        const std::string &g = strs[v.i]; const size_t n = strlen(suffix);
        return g.size() >= n && g.compare(g.size() - n, n, suffix) == 0;
    }
    int64_t find_mark() const { for (int64_t k = (int64_t)stack.size() - 1; k >= 0; k--) if (stack[k].tag == T_MARK) return k; return -1; }
    void memoize(int64_t idx, const Val &v) { memo[idx] = v; if (Box *b = box_of(v)) { if (b->memo_id < 0) b->memo_id = idx; } }

    bool append_items(Box &dst, size_t from) {             // list.extend(stack[from:])
        for (size_t k = from; k < stack.size(); k++) {
            const Val &v = stack[k];
            switch (dst.role) {
            case R_SINK_STATES:
                if (v.tag != T_STATE || v.i != dst.count) return fail("`states` holds something that is not the next SAMState in order");
                dst.count++; break;
            case R_SINK_DISCARD: free_val(v); dst.count++; break;
            case R_SINK_TEXT:
                if (v.tag != T_INT || v.i < INT32_MIN || v.i > INT32_MAX) return fail("`input_ids` holds a non-int32 element");
                text.push_back((int32_t)v.i); dst.count++; break;
            default: dst.items.push_back(v);
            }
        }
        stack.resize(from);
        return true;
    }
    bool set_items(Box &d, size_t from) {                   // dict.update(pairs on the stack)
        if ((stack.size() - from) & 1) return fail("odd number of dict items");
        for (size_t k = from; k < stack.size(); k++) d.items.push_back(stack[k]);
        stack.resize(from);
        return true;
    }
    // BUILD of a SAMState: its __dict__ -> one row of the tables, its `next` dict -> the edge arrays (dict order), everything freed
    bool sink_state(const Val &state) {
        if (state.tag != T_DICT || !box_of(state)) return fail("SAMState state is not a dict");
        Box &d = boxes[state.i];
        int64_t lk = 0, len = 0, ax = 0; bool got_l = false, got_len = false, got_ax = false, got_next = false;
        for (size_t k = 0; k + 1 < d.items.size(); k += 2) {
            const Val &key = d.items[k], &v = d.items[k + 1];
            if (key.tag != T_STR) return fail("SAMState attribute name is not a string");
            const std::string &name = strs[key.i];
            if (name == "next") {
                if (v.tag != T_DICT || !box_of(v)) return fail("SAMState.next is not a dict");
                const Box &nx = boxes[v.i];
                if (nx.items.size() / 2 > (size_t)INT32_MAX) return fail("degree overflow");
                for (size_t e = 0; e + 1 < nx.items.size(); e += 2) {
                    const Val &t = nx.items[e], &dst = nx.items[e + 1];
                    if (t.tag != T_INT || dst.tag != T_INT || t.i < INT32_MIN || t.i > INT32_MAX || dst.i < 0 || dst.i > INT32_MAX) return fail("SAMState.next holds a non-int32 item");
                    etok.push_back((int32_t)t.i); edst.push_back((int32_t)dst.i);
                }
                deg.push_back((int32_t)(nx.items.size() / 2)); got_next = true;
            } else if (name == "link" || name == "length" || name == "cnt_endpos" || name == "min_endpos") {
                if (v.tag != T_INT || v.i < INT32_MIN || v.i > INT32_MAX) return fail("SAMState." + name + " is not an int32");
                if (name == "link") { lk = v.i; got_l = true; }
                else if (name == "length") { len = v.i; got_len = true; }
                else {
                    const int is_me = name == "min_endpos";
                    if (aux_is_minend >= 0 && aux_is_minend != is_me) return fail("states mix cnt_endpos and min_endpos");
                    aux_is_minend = is_me; ax = v.i; got_ax = true;
                }
            }                                               // (other attributes: ignored)
        }
        if (!got_l || !got_len || !got_ax || !got_next) return fail("SAMState lacks next / link / length / cnt_endpos|min_endpos");
        link.push_back((int32_t)lk); length.push_back((int32_t)len); aux.push_back((int32_t)ax);
        n_states++;
        free_val(state);
        return true;
    }
};

static inline int64_t le_int(const unsigned char *p, int n, bool sign) {
    uint64_t v = 0;
    for (int k = 0; k < n; k++) v |= (uint64_t)p[k] << (8 * k);
    if (sign && n < 8 && (p[n - 1] & 0x80)) v |= ~0ull << (8 * n);
    return (int64_t)v;
}

bool run(Reader &r, VM &vm, Val &result) {
    unsigned char op;
    unsigned char tmp[16];
    std::string s, s2;
    for (;;) {
        if (!r.byte(op)) return vm.fail("unexpected end of the pickle");
        switch (op) {
        case 0x80: if (!r.byte(tmp[0])) return vm.fail("truncated"); if (tmp[0] < 2 || tmp[0] > 5) return vm.fail("pickle protocol " + std::to_string(tmp[0]) + " is not supported (2-5)"); break;   // PROTO
        case 0x95: if (!r.bytes(tmp, 8)) return vm.fail("truncated"); break;                                          // FRAME: a length hint
        case '.': if (vm.stack.empty()) return vm.fail("STOP on an empty stack"); result = vm.stack.back(); return true;
        case '(': vm.stack.emplace_back(T_MARK, 0); break;
        case 'N': vm.stack.emplace_back(T_NONE, 0); break;
        case 0x88: vm.stack.emplace_back(T_BOOL, 1); break;
        case 0x89: vm.stack.emplace_back(T_BOOL, 0); break;
        case 'K': if (!r.bytes(tmp, 1)) return vm.fail("truncated"); vm.stack.emplace_back(T_INT, le_int(tmp, 1, false)); break;
        case 'M': if (!r.bytes(tmp, 2)) return vm.fail("truncated"); vm.stack.emplace_back(T_INT, le_int(tmp, 2, false)); break;
        case 'J': if (!r.bytes(tmp, 4)) return vm.fail("truncated"); vm.stack.emplace_back(T_INT, le_int(tmp, 4, true)); break;
        case 0x8a: {                                                                                                   // LONG1
            if (!r.bytes(tmp, 1)) return vm.fail("truncated");
            const int n = tmp[0];
            if (n > 8) return vm.fail("integer wider than 64 bits");
            if (n && !r.bytes(tmp, n)) return vm.fail("truncated");
            vm.stack.emplace_back(T_INT, n ? le_int(tmp, n, true) : 0); break;
        }
        case 'G': {                                                                                                    // BINFLOAT, big-endian
            if (!r.bytes(tmp, 8)) return vm.fail("truncated");
            uint64_t v = 0; for (int k = 0; k < 8; k++) v = (v << 8) | tmp[k];
            Val x; x.tag = T_FLOAT; memcpy(&x.f, &v, 8); vm.stack.push_back(x); break;
        }
        case 0x8c: case 'X': case 0x8d: {                                                                              // SHORT_BINUNICODE / BINUNICODE / BINUNICODE8
            const int w = op == 0x8c ? 1 : (op == 'X' ? 4 : 8);
            if (!r.bytes(tmp, w)) return vm.fail("truncated");
            const uint64_t n = (uint64_t)le_int(tmp, w, false);
            if (n > (1u << 20)) { if (!r.skip(n)) return vm.fail("truncated"); vm.stack.emplace_back(T_OPAQUE, 0); break; }
            s.resize((size_t)n);
            if (n && !r.bytes(&s[0], (size_t)n)) return vm.fail("truncated");
            vm.stack.emplace_back(T_STR, vm.str_id(std::string(s))); break;
        }
        case 'C': case 'B': case 0x8e: case 'U': case 'T': {                                                           // bytes / old strings: opaque
            const int w = (op == 'C' || op == 'U') ? 1 : (op == 0x8e ? 8 : 4);
            if (!r.bytes(tmp, w)) return vm.fail("truncated");
            if (!r.skip((uint64_t)le_int(tmp, w, false))) return vm.fail("truncated");
            vm.stack.emplace_back(T_OPAQUE, 0); break;
        }
        case '}': vm.stack.emplace_back(T_DICT, vm.new_box()); break;
        case ')': vm.stack.emplace_back(T_TUPLE, vm.new_box()); break;
        case ']': {
            Role role = R_NORMAL;
            if (!vm.stack.empty()) {                                                   // the key this list will be stored under precedes it on the stack
                const Val &k = vm.stack.back();
                if (vm.str_is(k, "states")) role = R_SINK_STATES;
                else if (vm.str_is(k, "states_topk_next")) role = R_SINK_DISCARD;
                else if (vm.str_is(k, "input_ids")) { role = R_SINK_TEXT; vm.have_text = true; }
            }
            vm.stack.emplace_back(T_LIST, vm.new_box(role)); break;
        }
        case 0x85: case 0x86: case 0x87: case 't': {                                                                   // TUPLE1/2/3, TUPLE
            size_t from;
            if (op == 't') { const int64_t m = vm.find_mark(); if (m < 0) return vm.fail("TUPLE without MARK"); from = (size_t)m + 1; }
            else { const size_t n = op - 0x84; if (vm.stack.size() < n) return vm.fail("stack underflow"); from = vm.stack.size() - n; }
            const int64_t id = vm.new_box();
            for (size_t k = from; k < vm.stack.size(); k++) vm.boxes[id].items.push_back(vm.stack[k]);
            vm.stack.resize(op == 't' ? from - 1 : from);
            vm.stack.emplace_back(T_TUPLE, id); break;
        }
        case 'a': case 'e': {                                                                                          // APPEND / APPENDS
            size_t from;
            if (op == 'e') { const int64_t m = vm.find_mark(); if (m < 1) return vm.fail("APPENDS without MARK"); from = (size_t)m + 1; }
            else { if (vm.stack.size() < 2) return vm.fail("stack underflow"); from = vm.stack.size() - 1; }
            const Val lst = vm.stack[op == 'e' ? from - 2 : from - 1];
            if (lst.tag != T_LIST || !vm.box_of(lst)) return vm.fail("APPEND to a non-list");
            if (!vm.append_items(vm.boxes[lst.i], from)) return false;
            if (op == 'e') vm.stack.pop_back();                                        // the MARK
            break;
        }
        case 's': case 'u': {                                                                                          // SETITEM / SETITEMS
            size_t from;
            if (op == 'u') { const int64_t m = vm.find_mark(); if (m < 1) return vm.fail("SETITEMS without MARK"); from = (size_t)m + 1; }
            else { if (vm.stack.size() < 3) return vm.fail("stack underflow"); from = vm.stack.size() - 2; }
            const Val d = vm.stack[op == 'u' ? from - 2 : from - 1];
            if (d.tag != T_DICT || !vm.box_of(d)) return vm.fail("SETITEM on a non-dict");
            if (!vm.set_items(vm.boxes[d.i], from)) return false;
            if (op == 'u') vm.stack.pop_back();
            break;
        }
        case 'c': {                                                                                                    // GLOBAL module\nname\n
            if (!r.line(s) || !r.line(s2)) return vm.fail("truncated GLOBAL");
            vm.stack.emplace_back(T_GLOBAL, vm.str_id(s + "\n" + s2)); break;
        }
        case 0x93: {                                                                                                   // STACK_GLOBAL
            if (vm.stack.size() < 2) return vm.fail("stack underflow");
            const Val name = vm.stack.back(), mod = vm.stack[vm.stack.size() - 2];
            if (name.tag != T_STR || mod.tag != T_STR) return vm.fail("STACK_GLOBAL needs two strings");
            vm.stack.resize(vm.stack.size() - 2);
            vm.stack.emplace_back(T_GLOBAL, vm.str_id(vm.strs[mod.i] + "\n" + vm.strs[name.i])); break;
        }
        case 0x81: case 'R': {                                                                                         // NEWOBJ / REDUCE: an inert record {callable, state}
            if (vm.stack.size() < 2) return vm.fail("stack underflow");
            const Val args = vm.stack.back(), cls = vm.stack[vm.stack.size() - 2];
            vm.stack.resize(vm.stack.size() - 2);
            // protocols 2 / 3 name a NESTED class (StaticSAM.SAMState) as getattr(<global StaticSAM>, 'SAMState'): still just a name
            if (op == 'R' && (vm.global_ends_with(cls, "\ngetattr")) && args.tag == T_TUPLE && vm.box_of(args, 2) && vm.boxes[args.i].items.size() == 2 &&
                vm.boxes[args.i].items[0].tag == T_GLOBAL && vm.boxes[args.i].items[1].tag == T_STR) {
                const std::string name = vm.strs[vm.boxes[args.i].items[0].i] + "." + vm.strs[vm.boxes[args.i].items[1].i];
                vm.free_val(args);
                vm.stack.emplace_back(T_GLOBAL, vm.str_id(std::string(name)));
                break;
            }
            vm.free_val(args);
            const int64_t id = vm.new_box();
            vm.boxes[id].items.push_back(cls); vm.boxes[id].items.emplace_back(T_NONE, 0);
            vm.stack.emplace_back(T_OBJ, id); break;
        }
        case 'b': {                                                                                                    // BUILD
            if (vm.stack.size() < 2) return vm.fail("stack underflow");
            const Val state = vm.stack.back(); vm.stack.pop_back();
            Val &obj = vm.stack.back();
            if (obj.tag != T_OBJ || !vm.box_of(obj, 2)) return vm.fail("BUILD on a non-object");
            Box &ob = vm.boxes[obj.i];
            if (vm.global_ends_with(ob.items[0], "StaticSAM.SAMState") || vm.global_ends_with(ob.items[0], "\nSAMState")) {
                if (!vm.sink_state(state)) return false;
                const int64_t memo_id = ob.memo_id;
                ob.memo_id = -1;
                vm.free_val(obj);
                obj = Val(T_STATE, vm.n_states - 1);
                if (memo_id >= 0) vm.memo[memo_id] = obj;
            } else {
                vm.free_val(ob.items[1]);
                ob.items[1] = state;
            }
            break;
        }
        case 'q': case 'r': case 0x94: {                                                                               // BINPUT / LONG_BINPUT / MEMOIZE
            if (vm.stack.empty()) return vm.fail("memoize on an empty stack");
            int64_t idx;
            if (op == 0x94) idx = vm.memo_next++;
            else { const int w = op == 'q' ? 1 : 4; if (!r.bytes(tmp, w)) return vm.fail("truncated"); idx = le_int(tmp, w, false); if (idx >= vm.memo_next) vm.memo_next = idx + 1; }
            vm.memoize(idx, vm.stack.back()); break;
        }
        case 'h': case 'j': {                                                                                          // BINGET / LONG_BINGET
            const int w = op == 'h' ? 1 : 4;
            if (!r.bytes(tmp, w)) return vm.fail("truncated");
            const auto it = vm.memo.find(le_int(tmp, w, false));
            if (it == vm.memo.end()) return vm.fail("GET of an object this reader already consumed (shared sub-objects are outside the supported subset)");
            // a second reference to a container / object would outlive the owner that frees it (see the ownership note in the header)
            if (VM::is_box(it->second)) return vm.fail("GET of a container or object (shared sub-objects are outside the supported subset)");
            vm.stack.push_back(it->second); break;
        }
        case '0': if (vm.stack.empty()) return vm.fail("stack underflow"); vm.stack.pop_back(); break;                  // POP
        case '1': { const int64_t m = vm.find_mark(); if (m < 0) return vm.fail("POP_MARK without MARK"); vm.stack.resize((size_t)m); break; }
        default: {
            char b[64]; snprintf(b, sizeof b, "pickle opcode 0x%02x is outside the supported subset", op);
            return vm.fail(b);
        }
        }
    }
}

}  // namespace

extern "C" int samd_static_from_pickle(const char *path, int32_t kind, double out_params[8], samd_static_t **out) {
    if (!path || !out || (kind != SAMD_KIND_COUNT && kind != SAMD_KIND_ENDPOS)) { samd_set_error("samd_static_from_pickle: invalid argument"); return SAMD_E_INVALID; }
    FILE *f = fopen(path, "rb");
    if (!f) { samd_set_error("cannot open %s", path); return SAMD_E_IO; }
    int rc = SAMD_OK;
    try {
        Reader r(f);
        VM vm;
        Val top;
        if (!run(r, vm, top)) { fclose(f); samd_set_error("%s: %s", path, vm.err.c_str()); return SAMD_E_IO; }
        fclose(f); f = nullptr;
        if (top.tag != T_OBJ || !vm.box_of(top, 2) || !vm.global_ends_with(vm.boxes[top.i].items[0], "\nStaticSAM")) { samd_set_error("%s: the pickle does not hold a StaticSAM object", path); return SAMD_E_IO; }
        const Val st = vm.boxes[top.i].items[1];
        if (st.tag != T_DICT || !vm.box_of(st)) { samd_set_error("%s: StaticSAM without attributes", path); return SAMD_E_IO; }
        if (out_params) for (int k = 0; k < 8; k++) out_params[k] = -1.0;
        int64_t listed = -1;
        const Box &d = vm.boxes[st.i];
        for (size_t k = 0; k + 1 < d.items.size(); k += 2) {
            const Val &key = d.items[k], &v = d.items[k + 1];
            if (key.tag != T_STR) continue;
            const std::string &name = vm.strs[key.i];
            if (name == "states") { if (v.tag == T_LIST && vm.box_of(v) && vm.boxes[v.i].role == R_SINK_STATES) listed = vm.boxes[v.i].count; continue; }
            if (!out_params) continue;
            const double num = v.tag == T_INT ? (double)v.i : (v.tag == T_FLOAT ? v.f : -1.0);
            // out_params: [0] max_predicts [1] alpha [2] K [3] n_predicts [4] cur_index [5] cur_length [6] last [7] max_length
            static const char *names[8] = {"max_predicts", "alpha", "K", "n_predicts", "cur_index", "cur_length", "last", "max_length"};
            for (int p = 0; p < 8; p++) if (name == names[p]) out_params[p] = num;
        }
        if (listed != vm.n_states || vm.n_states < 1) { samd_set_error("%s: `states` lists %lld states, %lld were read", path, (long long)listed, (long long)vm.n_states); return SAMD_E_IO; }
        if ((kind == SAMD_KIND_ENDPOS) != (vm.aux_is_minend == 1)) {
            samd_set_error("%s: the states carry %s, the caller asked for the other variant", path, vm.aux_is_minend == 1 ? "min_endpos (samd)" : "cnt_endpos (samd_sam_only)");
            return SAMD_E_IO;
        }
        if (kind == SAMD_KIND_ENDPOS && !vm.have_text) { samd_set_error("%s: no input_ids in a samd (min_endpos) automaton", path); return SAMD_E_IO; }
        // edge targets are checked against the state count by from_tables' layout; release what is no longer needed first
        std::vector<Box>().swap(vm.boxes);
        rc = samd_static_from_tables(kind, vm.n_states, vm.link.data(), vm.length.data(), vm.aux.data(), vm.deg.data(), vm.etok.data(), vm.edst.data(),
                                     kind == SAMD_KIND_ENDPOS ? vm.text.data() : nullptr, kind == SAMD_KIND_ENDPOS ? (int64_t)vm.text.size() : 0, out);
    } catch (const std::exception &e) {
        if (f) fclose(f);
        samd_set_error("%s: %s", path, e.what());
        return SAMD_E_CAPACITY;
    }
    return rc;
}
