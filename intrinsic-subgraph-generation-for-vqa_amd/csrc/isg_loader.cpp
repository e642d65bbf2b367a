// Host-side scene-graph loader (include/isg_loader.h): a single-pass JSON reader that converts every image's scene
// graph to the reference's token tensors while parsing, and a collate that writes a PyG-style batch straight into
// caller-owned (pinned) buffers.  Plain C++17; no GPU, no third-party JSON library.
//
// Reference behaviour: ISubGVQA/datasets/scene_graph.py:145-389, ISubGVQA/datasets/gqa.py:170-175,258 (see the header).
// The conversion keeps the reference's order of everything that reaches a tensor: objects by sorted id string, per
// object its self loop, then its relations in file order, each followed by the reverse edge when the reverse pair is
// absent from the file.  Attributes: first-occurrence order of the distinct strings (the reference iterates a Python
// set, whose order is hash-seed dependent).
#include "../../include/isg_loader.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <string_view>
#include <thread>
#include <unordered_map>
#include <vector>

namespace {

thread_local std::string g_err;
int fail(int code, std::string msg) {
  g_err = std::move(msg);
  return code;
}

constexpr int64_t TOK_UNK = 0, TOK_PAD = 1, TOK_SELF = 4;   // positions of the specials (scene_graph.py:172-178)

}  // namespace

struct isg_sg_vocab {
  std::unordered_map<std::string, int64_t> stoi;
  int64_t lookup(std::string_view t) const {
    auto it = stoi.find(std::string(t));
    return it == stoi.end() ? -1 : it->second;
  }
  // vocab_sg.get_stoi().get(token, 1): out-of-vocabulary maps to 1 = <pad> (scene_graph.py:287,298,328)
  int64_t get(std::string_view t) const {
    const int64_t i = lookup(t);
    return i < 0 ? TOK_PAD : i;
  }
};

namespace {

// ---- JSON ---------------------------------------------------------------------------------------------------------
struct Parser {
  const char *p, *end, *begin;
  std::string err;

  bool bad(const char *what) {
    if (err.empty()) err = std::string(what) + " at byte " + std::to_string(p - begin);
    return false;
  }
  void ws() {
    while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p;
  }
  bool eat(char c) {
    ws();
    if (p < end && *p == c) { ++p; return true; }
    return false;
  }
  bool expect(char c) { return eat(c) ? true : bad((std::string("expected '") + c + "'").c_str()); }

  static void utf8(std::string &out, unsigned cp) {
    if (cp < 0x80) out += (char)cp;
    else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
    else if (cp < 0x10000) { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
    else { out += (char)(0xF0 | (cp >> 18)); out += (char)(0x80 | ((cp >> 12) & 0x3F)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
  }
  bool hex4(unsigned &v) {
    if (end - p < 4) return bad("truncated \\u escape");
    v = 0;
    for (int i = 0; i < 4; ++i) {
      const char c = *p++;
      v <<= 4;
      if (c >= '0' && c <= '9') v |= c - '0';
      else if (c >= 'a' && c <= 'f') v |= c - 'a' + 10;
      else if (c >= 'A' && c <= 'F') v |= c - 'A' + 10;
      else return bad("bad \\u escape");
    }
    return true;
  }
  // A string; `view` points into the input when it has no escapes, into `scratch` otherwise.
  bool str(std::string_view &view, std::string &scratch) {
    ws();
    if (p >= end || *p != '"') return bad("expected string");
    const char *s = ++p;
    while (p < end && *p != '"' && *p != '\\') ++p;
    if (p >= end) return bad("unterminated string");
    if (*p == '"') { view = std::string_view(s, p - s); ++p; return true; }
    scratch.assign(s, p - s);
    while (p < end && *p != '"') {
      if (*p != '\\') { scratch += *p++; continue; }
      if (++p >= end) return bad("unterminated escape");
      const char c = *p++;
      switch (c) {
        case '"': scratch += '"'; break;
        case '\\': scratch += '\\'; break;
        case '/': scratch += '/'; break;
        case 'b': scratch += '\b'; break;
        case 'f': scratch += '\f'; break;
        case 'n': scratch += '\n'; break;
        case 'r': scratch += '\r'; break;
        case 't': scratch += '\t'; break;
        case 'u': {
          unsigned cp = 0;
          if (!hex4(cp)) return false;
          if (cp >= 0xD800 && cp < 0xDC00 && end - p >= 6 && p[0] == '\\' && p[1] == 'u') {   // surrogate pair
            p += 2;
            unsigned lo = 0;
            if (!hex4(lo)) return false;
            cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
          }
          utf8(scratch, cp);
          break;
        }
        default: return bad("bad escape");
      }
    }
    if (p >= end) return bad("unterminated string");
    ++p;
    view = scratch;
    return true;
  }
  bool number(double &v) {
    ws();
    // the text is a (pointer, length) view with no terminator: scan the number's own characters inside [p, end) and
    // convert from a bounded, NUL-terminated copy (strtod on the raw pointer could read past the buffer)
    char buf[64];
    size_t n = 0;
    while (p + n < end && n + 1 < sizeof(buf)) {
      const char c = p[n];
      if (!((c >= '0' && c <= '9') || c == '-' || c == '+' || c == '.' || c == 'e' || c == 'E')) break;
      buf[n++] = c;
    }
    buf[n] = 0;
    char *e = nullptr;
    v = std::strtod(buf, &e);
    if (e == buf) return bad("expected number");
    p += e - buf;
    return true;
  }
  int depth = 0;                      // nesting of skipped containers
  static constexpr int MAX_DEPTH = 64; // scene-graph JSON nests 4 deep; a hostile file must not overflow the stack
  bool skip() {   // any value
    ws();
    if (p >= end) return bad("unexpected end");
    if (depth >= MAX_DEPTH) return bad("nesting too deep");
    struct Guard { int &d; Guard(int &x) : d(x) { ++d; } ~Guard() { --d; } } guard(depth);
    std::string_view v;
    std::string tmp;
    switch (*p) {
      case '"': return str(v, tmp);
      case '{':
        ++p;
        if (eat('}')) return true;
        do {
          if (!str(v, tmp) || !expect(':') || !skip()) return false;
        } while (eat(','));
        return expect('}');
      case '[':
        ++p;
        if (eat(']')) return true;
        do {
          if (!skip()) return false;
        } while (eat(','));
        return expect(']');
      case 't': if (end - p >= 4 && !memcmp(p, "true", 4)) { p += 4; return true; } return bad("bad literal");
      case 'f': if (end - p >= 5 && !memcmp(p, "false", 5)) { p += 5; return true; } return bad("bad literal");
      case 'n': if (end - p >= 4 && !memcmp(p, "null", 4)) { p += 4; return true; } return bad("bad literal");
      default: { double d; return number(d); }
    }
  }
};

// ---- one image's objects while parsing ---------------------------------------------------------------------------------
struct Rel {
  std::string target;   // object id
  int64_t tok;
};
struct Obj {
  std::string id;
  int64_t name_tok = -1;
  bool has_name = false, has_attrs = false, has_rels = false;
  int64_t attr[3] = {TOK_PAD, TOK_PAD, TOK_PAD};
  int64_t bbox[4] = {-1, -1, -1, -1};
  std::vector<Rel> rels;
};

struct Graph {   // converted tensors of one image, offsets into the store's pools
  int64_t node0 = 0, n = 0, edge0 = 0, e = 0, sym0 = 0, s = 0;
};

}  // namespace

struct isg_sg_store {
  const isg_sg_vocab *vocab = nullptr;
  std::vector<int64_t> x, bbox;          // [*, 4]
  std::vector<int32_t> src, dst, sym;    // per-graph local ids / edge positions
  std::vector<int64_t> ea;
  std::vector<Graph> graphs;
  std::unordered_map<std::string, int64_t> slot;
  Graph missing;                          // query_and_translate's 6-node dummy

  // convert_one_gqa_scene_graph (scene_graph.py:231-389) on already tokenised objects
  int convert(std::vector<Obj> &objs, Graph &g) {
    std::sort(objs.begin(), objs.end(), [](const Obj &a, const Obj &b) { return a.id < b.id; });   // :234
    const int n = (int)objs.size();
    auto node_of = [&](const std::string &id) -> int {
      auto it = std::lower_bound(objs.begin(), objs.end(), id, [](const Obj &o, const std::string &k) { return o.id < k; });
      return (it != objs.end() && it->id == id) ? (int)(it - objs.begin()) : -1;
    };
    std::vector<uint64_t> pairs;          // from_to_connections_set (:255-263)
    std::vector<int> tgt;
    for (int i = 0; i < n; ++i)
      for (const Rel &r : objs[i].rels) {
        const int j = node_of(r.target);
        if (j < 0) return fail(ISG_LD_EPARSE, "relation of object '" + objs[i].id + "' points to unknown object '" + r.target + "'");
        tgt.push_back(j);
        pairs.push_back(((uint64_t)i << 32) | (uint32_t)j);
      }
    std::sort(pairs.begin(), pairs.end());
    g.node0 = (int64_t)x.size() / 4;
    g.edge0 = (int64_t)ea.size();
    g.sym0 = (int64_t)sym.size();
    size_t t = 0;
    for (int i = 0; i < n; ++i) {
      const Obj &o = objs[i];
      x.push_back(o.name_tok);
      x.insert(x.end(), o.attr, o.attr + 3);
      bbox.insert(bbox.end(), o.bbox, o.bbox + 4);
      src.push_back(i); dst.push_back(i); ea.push_back(TOK_SELF);                     // :311-315 self loop first
      for (const Rel &r : o.rels) {
        const int j = tgt[t++];
        src.push_back(i); dst.push_back(j); ea.push_back(r.tok);                      // :317-328
        if (!std::binary_search(pairs.begin(), pairs.end(), ((uint64_t)j << 32) | (uint32_t)i)) {   // :331-345
          src.push_back(j); dst.push_back(i); ea.push_back(r.tok);
          sym.push_back((int32_t)((int64_t)ea.size() - g.edge0 - 1));
        }
      }
    }
    g.n = n;
    g.e = (int64_t)ea.size() - g.edge0;
    g.s = (int64_t)sym.size() - g.sym0;
    return ISG_LD_OK;
  }

  int add_dummy(const int *targets, int n, Graph &g) {   // "<unk>" objects "0".."n-1", one relation each
    std::vector<Obj> objs(n);
    const int64_t unk = vocab->get("<unk>");
    for (int i = 0; i < n; ++i) {
      objs[i].id = std::to_string(i);
      objs[i].name_tok = unk;
      objs[i].attr[0] = unk;
      objs[i].rels.push_back({std::to_string(targets[i]), unk});
    }
    return convert(objs, g);
  }

  int parse_object(Parser &ps, Obj &o) {
    std::string_view key, val;
    std::string kt, vt;
    if (!ps.expect('{')) return ISG_LD_EPARSE;
    if (ps.eat('}')) return ISG_LD_OK;
    do {
      if (!ps.str(key, kt) || !ps.expect(':')) return ISG_LD_EPARSE;
      if (key == "name") {
        if (!ps.str(val, vt)) return ISG_LD_EPARSE;
        o.name_tok = vocab->get(val);                                                 // :286-287
        o.has_name = true;
      } else if (key == "attributes") {
        o.has_attrs = true;
        if (!ps.expect('[')) return ISG_LD_EPARSE;
        std::vector<std::string> seen;
        if (!ps.eat(']')) {
          do {
            if (!ps.str(val, vt)) return ISG_LD_EPARSE;
            if (std::find(seen.begin(), seen.end(), val) == seen.end()) seen.emplace_back(val);
          } while (ps.eat(','));
          if (!ps.expect(']')) return ISG_LD_EPARSE;
        }
        for (size_t a = 0; a < seen.size() && a < 3; ++a) o.attr[a] = vocab->get(seen[a]);   // :293-299
      } else if (key == "relations") {
        o.has_rels = true;
        if (!ps.expect('[')) return ISG_LD_EPARSE;
        if (!ps.eat(']')) {
          do {
            Rel r;
            bool has_obj = false, has_nm = false;
            if (!ps.expect('{')) return ISG_LD_EPARSE;
            if (!ps.eat('}')) {
              do {
                if (!ps.str(key, kt) || !ps.expect(':')) return ISG_LD_EPARSE;
                if (key == "object") { if (!ps.str(val, vt)) return ISG_LD_EPARSE; r.target = std::string(val); has_obj = true; }
                else if (key == "name") { if (!ps.str(val, vt)) return ISG_LD_EPARSE; r.tok = vocab->get(val); has_nm = true; }
                else if (!ps.skip()) return ISG_LD_EPARSE;
              } while (ps.eat(','));
              if (!ps.expect('}')) return ISG_LD_EPARSE;
            }
            if (!has_obj || !has_nm) { ps.bad("relation without 'object' or 'name'"); return ISG_LD_EPARSE; }
            o.rels.push_back(std::move(r));
          } while (ps.eat(','));
          if (!ps.expect(']')) return ISG_LD_EPARSE;
        }
      } else if (key.size() == 2 && (key[0] == 'x' || key[0] == 'y') && (key[1] == '1' || key[1] == '2')) {
        double d;
        if (!ps.number(d)) return ISG_LD_EPARSE;
        o.bbox[(key[0] == 'y' ? 1 : 0) + (key[1] == '2' ? 2 : 0)] = (int64_t)d;       // :301-306
      } else if (!ps.skip()) {
        return ISG_LD_EPARSE;
      }
    } while (ps.eat(','));
    return ps.expect('}') ? ISG_LD_OK : ISG_LD_EPARSE;
  }

  int add_json(const char *text, int64_t len) {
    Parser ps{text, text + len, text, {}};
    std::string_view key, img;
    std::string kt, it;
    std::vector<Obj> objs;
    auto perr = [&](int code) { return fail(code, "scene-graph JSON: " + (ps.err.empty() ? g_err : ps.err)); };
    if (!ps.expect('{')) return perr(ISG_LD_EPARSE);
    if (!ps.eat('}')) {
      do {
        if (!ps.str(img, it) || !ps.expect(':') || !ps.expect('{')) return perr(ISG_LD_EPARSE);
        const std::string image_id(img);
        objs.clear();
        bool has_objects = false;
        if (!ps.eat('}')) {
          do {
            if (!ps.str(key, kt) || !ps.expect(':')) return perr(ISG_LD_EPARSE);
            if (key == "objects") {
              has_objects = true;
              if (!ps.expect('{')) return perr(ISG_LD_EPARSE);
              if (!ps.eat('}')) {
                do {
                  Obj o;
                  std::string_view oid;
                  std::string ot;
                  if (!ps.str(oid, ot) || !ps.expect(':')) return perr(ISG_LD_EPARSE);
                  o.id = std::string(oid);
                  if (parse_object(ps, o) != ISG_LD_OK) return perr(ISG_LD_EPARSE);
                  if (!o.has_name || !o.has_attrs || !o.has_rels) {
                    ps.bad(("object '" + o.id + "' of image '" + image_id + "' lacks name/attributes/relations").c_str());
                    return perr(ISG_LD_EPARSE);
                  }
                  // a repeated key keeps the last value, like json.load
                  auto dup = std::find_if(objs.begin(), objs.end(), [&](const Obj &q) { return q.id == o.id; });
                  if (dup != objs.end()) *dup = std::move(o); else objs.push_back(std::move(o));
                } while (ps.eat(','));
                if (!ps.expect('}')) return perr(ISG_LD_EPARSE);
              }
            } else if (!ps.skip()) {
              return perr(ISG_LD_EPARSE);
            }
          } while (ps.eat(','));
          if (!ps.expect('}')) return perr(ISG_LD_EPARSE);
        }
        if (!has_objects) { ps.bad(("image '" + image_id + "' has no 'objects'").c_str()); return perr(ISG_LD_EPARSE); }
        Graph g;
        int rc;
        if (objs.empty()) {                                                           // :202-229 two-node dummy
          static const int two[2] = {1, 0};
          rc = add_dummy(two, 2, g);
        } else {
          rc = convert(objs, g);
        }
        if (rc != ISG_LD_OK) { g_err = "image '" + image_id + "': " + g_err; return rc; }
        auto found = slot.find(image_id);
        if (found == slot.end()) { slot.emplace(image_id, (int64_t)graphs.size()); graphs.push_back(g); }
        else graphs[found->second] = g;                                               // dict `|`: the later file wins
      } while (ps.eat(','));
      if (!ps.expect('}')) return perr(ISG_LD_EPARSE);
    }
    ps.ws();
    if (ps.p != ps.end) { ps.bad("trailing data"); return perr(ISG_LD_EPARSE); }
    return ISG_LD_OK;
  }

  // query_and_translate (:138-142): unknown id, or a graph whose only edge is one self loop -> the 6-node dummy
  const Graph &query(int64_t slot_id) const {
    if (slot_id < 0 || slot_id >= (int64_t)graphs.size()) return missing;
    const Graph &g = graphs[slot_id];
    return g.e == 1 ? missing : g;
  }
};

extern "C" {

int isg_loader_abi_version(void) { return ISG_LOADER_ABI_VERSION; }
const char *isg_sg_last_error(void) { return g_err.c_str(); }

int isg_sg_vocab_build(const char *const *tokens, int64_t n_tokens, isg_sg_vocab **out) {
  if (!out || n_tokens < 0 || (n_tokens > 0 && !tokens)) return fail(ISG_LD_EINVAL, "isg_sg_vocab_build: bad arguments");
  std::vector<std::string> flat;
  flat.reserve((size_t)n_tokens + 2);
  for (int64_t i = 0; i < n_tokens; ++i) {
    if (!tokens[i]) return fail(ISG_LD_EINVAL, "isg_sg_vocab_build: NULL token");
    flat.emplace_back(tokens[i]);
  }
  flat.emplace_back("<self>");     // scene_graph.py:164
  flat.emplace_back("pokemon");    // :165
  // {token: position}: the dict keeps first-insertion ORDER and the last POSITION (:167)
  std::unordered_map<std::string, int64_t> last;
  std::vector<const std::string *> order;
  for (size_t i = 0; i < flat.size(); ++i) {
    auto ins = last.emplace(flat[i], (int64_t)i);
    if (ins.second) order.push_back(&ins.first->first); else ins.first->second = (int64_t)i;
  }
  static const char *specials[5] = {"<unk>", "<pad>", "<sos>", "<eos>", "<self>"};
  auto *v = new isg_sg_vocab;
  for (int i = 0; i < 5; ++i) v->stoi.emplace(specials[i], i);                        // special_first
  int64_t next = 5;
  for (const std::string *tok : order) {
    if (v->stoi.count(*tok)) continue;                                                // a special: already placed
    if (last[*tok] < 1) continue;                                                     // 'frequency' 0 < min_freq = 1
    v->stoi.emplace(*tok, next++);
  }
  *out = v;
  return ISG_LD_OK;
}

int64_t isg_sg_vocab_size(const isg_sg_vocab *v) { return v ? (int64_t)v->stoi.size() : -1; }
int64_t isg_sg_vocab_lookup(const isg_sg_vocab *v, const char *token) { return (v && token) ? v->lookup(token) : -1; }
void isg_sg_vocab_free(isg_sg_vocab *v) { delete v; }

int isg_sg_store_create(const isg_sg_vocab *v, isg_sg_store **out) {
  if (!v || !out) return fail(ISG_LD_EINVAL, "isg_sg_store_create: bad arguments");
  auto *s = new isg_sg_store;
  s->vocab = v;
  static const int six[6] = {1, 0, 3, 1, 5, 3};                                       // scene_graph.py:72-137
  const int rc = s->add_dummy(six, 6, s->missing);
  if (rc != ISG_LD_OK) { delete s; return rc; }
  *out = s;
  return ISG_LD_OK;
}

int isg_sg_store_add_json(isg_sg_store *s, const char *text, int64_t len) {
  if (!s || !text || len < 0) return fail(ISG_LD_EINVAL, "isg_sg_store_add_json: bad arguments");
  return s->add_json(text, len);
}

int isg_sg_store_add_json_file(isg_sg_store *s, const char *path) {
  if (!s || !path) return fail(ISG_LD_EINVAL, "isg_sg_store_add_json_file: bad arguments");
  FILE *f = std::fopen(path, "rb");
  if (!f) return fail(ISG_LD_EIO, std::string("cannot open ") + path);
  std::string buf;
  std::fseek(f, 0, SEEK_END);
  const long size = std::ftell(f);
  std::fseek(f, 0, SEEK_SET);
  if (size < 0) { std::fclose(f); return fail(ISG_LD_EIO, std::string("cannot size ") + path); }
  buf.resize((size_t)size);
  const size_t got = size ? std::fread(&buf[0], 1, (size_t)size, f) : 0;
  std::fclose(f);
  if (got != (size_t)size) return fail(ISG_LD_EIO, std::string("short read on ") + path);
  return s->add_json(buf.data(), (int64_t)buf.size());
}

int64_t isg_sg_store_num_graphs(const isg_sg_store *s) { return s ? (int64_t)s->graphs.size() : -1; }

int64_t isg_sg_store_find(const isg_sg_store *s, const char *image_id) {
  if (!s || !image_id) return -1;
  auto it = s->slot.find(image_id);
  return it == s->slot.end() ? -1 : it->second;
}

void isg_sg_store_free(isg_sg_store *s) { delete s; }

int isg_sg_store_find_many(const isg_sg_store *s, const char *const *image_ids, int64_t B, int64_t *slots) {
  if (!s || B < 0 || (B > 0 && (!image_ids || !slots))) return fail(ISG_LD_EINVAL, "isg_sg_store_find_many: bad arguments");
  for (int64_t b = 0; b < B; ++b) {
    if (!image_ids[b]) return fail(ISG_LD_EINVAL, "isg_sg_store_find_many: NULL image id");
    auto it = s->slot.find(image_ids[b]);
    slots[b] = it == s->slot.end() ? -1 : it->second;
  }
  return ISG_LD_OK;
}

int isg_sg_collate_sizes(const isg_sg_store *s, const int64_t *slots, int64_t B, int64_t *totals) {
  if (!s || B < 0 || (B > 0 && !slots) || !totals) return fail(ISG_LD_EINVAL, "isg_sg_collate_sizes: bad arguments");
  int64_t n = 0, e = 0, y = 0;
  for (int64_t b = 0; b < B; ++b) {
    const Graph &g = s->query(slots[b]);
    n += g.n; e += g.e; y += g.s;
  }
  totals[0] = n; totals[1] = e; totals[2] = y;
  return ISG_LD_OK;
}

int isg_sg_collate(const isg_sg_store *s, const int64_t *slots, int64_t B, int64_t *x, int64_t *edge_index,
                   int64_t *edge_attr, int64_t *x_bbox, int64_t *added_sym_edge, int64_t *batch, int64_t *ptr,
                   int64_t *bounds, int32_t n_threads) {
  int64_t tot[3];
  int rc = isg_sg_collate_sizes(s, slots, B, tot);
  if (rc != ISG_LD_OK) return rc;
  if (!ptr || (tot[0] > 0 && (!x || !x_bbox || !batch)) || (tot[1] > 0 && (!edge_index || !edge_attr)) ||
      (tot[2] > 0 && !added_sym_edge))
    return fail(ISG_LD_EINVAL, "isg_sg_collate: NULL output buffer");
  const int64_t E = tot[1];
  // pass 1: where every graph lands (node / edge / sym offsets), bounds
  std::vector<int64_t> e_off((size_t)B + 1), y_off((size_t)B + 1);
  int64_t nmax = 0, emax = 0;
  ptr[0] = 0; e_off[0] = 0; y_off[0] = 0;
  for (int64_t b = 0; b < B; ++b) {
    const Graph &g = s->query(slots[b]);
    ptr[b + 1] = ptr[b] + g.n;
    e_off[b + 1] = e_off[b] + g.e;
    y_off[b + 1] = y_off[b] + g.s;
    nmax = std::max(nmax, g.n);
    emax = std::max(emax, g.e);
  }
  // pass 2: graphs are independent; split them into contiguous ranges of roughly equal edge counts
  auto work = [&](int64_t b0, int64_t b1) {
    for (int64_t b = b0; b < b1; ++b) {
      const Graph &g = s->query(slots[b]);
      const int64_t n0 = ptr[b], e0 = e_off[b], y0 = y_off[b];
      std::memcpy(x + n0 * 4, s->x.data() + g.node0 * 4, (size_t)g.n * 4 * sizeof(int64_t));
      std::memcpy(x_bbox + n0 * 4, s->bbox.data() + g.node0 * 4, (size_t)g.n * 4 * sizeof(int64_t));
      std::memcpy(edge_attr + e0, s->ea.data() + g.edge0, (size_t)g.e * sizeof(int64_t));
      for (int64_t i = 0; i < g.n; ++i) batch[n0 + i] = b;
      const int32_t *gs = s->src.data() + g.edge0, *gd = s->dst.data() + g.edge0;
      int64_t *rs = edge_index + e0, *rd = edge_index + E + e0;
      for (int64_t t = 0; t < g.e; ++t) {                                              // edge_index += node offset
        rs[t] = n0 + gs[t];
        rd[t] = n0 + gd[t];
      }
      for (int64_t t = 0; t < g.s; ++t) added_sym_edge[y0 + t] = s->sym[g.sym0 + t];  // no offset (quirk Q6)
    }
  };
  const int T = (int)std::max<int64_t>(1, std::min<int64_t>(n_threads, B / 64));
  if (T <= 1) {
    work(0, B);
  } else {
    std::vector<int64_t> cut((size_t)T + 1, B);
    cut[0] = 0;
    for (int t = 1; t < T; ++t) {
      const int64_t at = std::lower_bound(e_off.begin(), e_off.end(), E * t / T) - e_off.begin();
      cut[t] = std::min<int64_t>(B, std::max<int64_t>(cut[t - 1], at));
    }
    std::vector<std::thread> pool;
    for (int t = 1; t < T; ++t)
      if (cut[t + 1] > cut[t]) pool.emplace_back(work, cut[t], cut[t + 1]);
    work(cut[0], cut[1]);
    for (auto &th : pool) th.join();
  }
  if (bounds) { bounds[0] = nmax; bounds[1] = emax; }
  return ISG_LD_OK;
}

}  // extern "C"
