// dw_cli.cc -- see dw_cli.h.  Plain C++17; talks to the sampler only through the C ABI.
#include "dw_cli.h"
#include "dw_multi.h"

#include "host_parallel.h"

#include <endian.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <memory>
#include <sstream>
#include <stdexcept>

namespace dw {

// ------------------------------------------------------------------ command line
namespace {
struct FlagSpec {
  const char *short_name, *long_name;
  bool takes_value;
};
// the `dw gibbs` flag table of src/cmd_parser.cc:59-113, plus this build's additions
const FlagSpec kFlags[] = {
    {"m", "fg_meta", true}, {"v", "variables", true}, {"", "domains", true},
    {"f", "factors", true}, {"w", "weights", true}, {"o", "outputFile", true},
    {"l", "n_learning_epoch", true}, {"i", "n_inference_epoch", true}, {"", "burn_in", true},
    {"c", "n_datacopy", true}, {"t", "n_threads", true}, {"a", "alpha", true},
    {"p", "stepsize", true}, {"d", "diminish", true}, {"b", "reg_param", true},
    {"", "regularization", true}, {"q", "quiet", false}, {"", "sample_evidence", false},
    {"", "learn_non_evidence", false}, {"", "noise_aware", false},
    {"", "device", true}, {"", "seed", true}, {"", "step_cap", true}, {"", "plan_layouts", true},
    {"", "gpus", true}, {"", "devices", true}, {"", "comm", true},
};

const FlagSpec *find_flag(const std::string &tok) {
  if (tok.size() >= 3 && tok[0] == '-' && tok[1] == '-') {
    for (const auto &f : kFlags) if (tok.substr(2) == f.long_name) return &f;
  } else if (tok.size() == 2 && tok[0] == '-') {
    for (const auto &f : kFlags) if (f.short_name[0] && tok[1] == f.short_name[0]) return &f;
  }
  return nullptr;
}

bool to_u64(const std::string &s, uint64_t &out) {
  if (s.empty() || s[0] == '-') return false;
  char *end = nullptr;
  errno = 0;
  unsigned long long v = strtoull(s.c_str(), &end, 10);
  if (errno || *end) return false;
  out = v;
  return true;
}
bool to_f64(const std::string &s, double &out) {
  if (s.empty()) return false;
  char *end = nullptr;
  errno = 0;
  double v = strtod(s.c_str(), &end);
  if (errno || *end) return false;
  out = v;
  return true;
}
}  // namespace

CmdLine parse_cmdline(int argc, const char *const argv[]) {
  CmdLine a;
  std::ostringstream err;
  a.app_name = argc > 1 ? argv[1] : "";
  if (a.app_name == "text2bin") {
    // dw text2bin MODE input output count_output [func_id arity var_is_positive...]
    // (positional, src/cmd_parser.cc:168-212)
    std::vector<std::string> pos(argv + 2, argv + argc);
    if (pos.size() < 4) {
      ++a.num_errors;
      err << "PARSE ERROR:\n             Required arguments missing: mode, input, output, count_output\n";
    } else {
      a.text2bin_mode = pos[0]; a.text2bin_input = pos[1]; a.text2bin_output = pos[2];
      a.text2bin_count_output = pos[3];
      if (a.text2bin_mode == "factor") {
        uint64_t u = 0;
        if (pos.size() < 7 || !to_u64(pos[4], u)) {
          ++a.num_errors;
          err << "PARSE ERROR:\n             factor needs: func_id arity var_is_positive...\n";
        } else {
          a.text2bin_factor_func_id = (int)u;
          if (!to_u64(pos[5], a.text2bin_factor_arity)) ++a.num_errors;
          for (size_t i = 6; i < pos.size(); ++i) {
            if (!to_u64(pos[i], u)) { ++a.num_errors; break; }
            a.text2bin_factor_variables_should_equal_to.push_back(u);
          }
        }
      }
    }
    a.error_text = err.str();
    return a;
  }
  if (a.app_name != "gibbs" && a.app_name != "bin2text") {
    ++a.num_errors;
    err << "DimmWitted (MI355X-native build)\nUsage: " << (argc ? argv[0] : "dw") << " MODE [ARG...]\n";
    for (const char *m : {"bin2text", "gibbs", "text2bin"}) err << "  " << (argc ? argv[0] : "dw") << " " << m << "\n";
    if (argc > 1) err << a.app_name << ": Unrecognized MODE\n";
    a.error_text = err.str();
    return a;
  }
  const bool is_gibbs = a.app_name == "gibbs";
  bool have_l = false, have_i = false;
  for (int i = 2; i < argc; ++i) {
    std::string tok = argv[i], val;
    // --flag=value
    size_t eq = tok.find('=');
    bool inline_val = false;
    if (tok.rfind("--", 0) == 0 && eq != std::string::npos) {
      val = tok.substr(eq + 1);
      tok = tok.substr(0, eq);
      inline_val = true;
    }
    const FlagSpec *f = find_flag(tok);
    if (!f) {
      ++a.num_errors;
      err << "PARSE ERROR: Argument: " << tok << "\n             Couldn't find match for argument\n";
      continue;
    }
    if (f->takes_value && !inline_val) {
      if (i + 1 >= argc) {
        ++a.num_errors;
        err << "PARSE ERROR: Argument: " << tok << "\n             Missing a value for this argument!\n";
        break;
      }
      val = argv[++i];
    }
    const std::string n = f->long_name;
    uint64_t u = 0;
    double d = 0;
    auto need_u = [&]() {
      if (!to_u64(val, u)) { ++a.num_errors; err << "PARSE ERROR: Argument: " << tok << "\n             Couldn't read argument value from string '" << val << "'\n"; return false; }
      return true;
    };
    auto need_d = [&]() {
      if (!to_f64(val, d)) { ++a.num_errors; err << "PARSE ERROR: Argument: " << tok << "\n             Couldn't read argument value from string '" << val << "'\n"; return false; }
      return true;
    };
    // multi-valued file flags accumulate; scalar flags: last value wins
    // (getLastValueOrDefault, src/cmd_parser.cc:16-21)
    if (n == "fg_meta") a.fg_file = val;
    else if (n == "variables") a.variable_file.push_back(val);
    else if (n == "domains") a.domain_file.push_back(val);
    else if (n == "factors") a.factor_file.push_back(val);
    else if (n == "weights") a.weight_file.push_back(val);
    else if (n == "outputFile") a.output_folder = val;
    else if (n == "n_learning_epoch") { if (need_u()) { a.n_learning_epoch = u; have_l = true; } }
    else if (n == "n_inference_epoch") { if (need_u()) { a.n_inference_epoch = u; have_i = true; } }
    else if (n == "burn_in") { if (need_u()) a.burn_in = u; }
    else if (n == "n_datacopy") { if (need_u()) a.n_datacopy = u; }
    else if (n == "n_threads") { if (need_u()) a.n_threads = u; }
    else if (n == "alpha") { if (need_d()) a.stepsize = d; }
    else if (n == "stepsize") { if (need_d()) a.stepsize2 = d; }
    else if (n == "diminish") { if (need_d()) a.decay = d; }
    else if (n == "reg_param") { if (need_d()) a.reg_param = d; }
    else if (n == "regularization") a.regularization_l1 = (val == "l1");
    else if (n == "quiet") a.should_be_quiet = true;
    else if (n == "sample_evidence") a.should_sample_evidence = true;
    else if (n == "learn_non_evidence") a.should_learn_non_evidence = true;
    else if (n == "noise_aware") a.is_noise_aware = true;
    else if (n == "device") { if (need_u()) a.device = (int)u; }
    else if (n == "seed") { if (need_u()) a.seed = u; }
    else if (n == "step_cap") { if (need_d()) a.step_cap = d; }
    else if (n == "plan_layouts") { if (need_u()) a.plan_layouts = (int)u; }
    else if (n == "gpus") { if (need_u()) a.gpus = (int)u; }
    else if (n == "comm") {
      if (val != "rccl" && val != "host") { ++a.num_errors; err << "PARSE ERROR: Argument: --comm\n             must be rccl or host\n"; }
      a.comm = val;
    } else if (n == "devices") {
      a.devices.clear();
      std::istringstream ss(val);
      std::string el;
      while (getline(ss, el, ',')) {
        if (!to_u64(el, u)) { ++a.num_errors; err << "PARSE ERROR: Argument: --devices\n             expects a comma-separated list of device ordinals\n"; break; }
        a.devices.push_back((int)u);
      }
    }
  }
  // -l and -i are required for gibbs (src/cmd_parser.cc:75-80)
  if (is_gibbs && !have_l) { ++a.num_errors; err << "PARSE ERROR:\n             Required argument missing: n_learning_epoch\n"; }
  if (is_gibbs && !have_i) { ++a.num_errors; err << "PARSE ERROR:\n             Required argument missing: n_inference_epoch\n"; }
  // XXX hack of the reference to support two step-size flags (src/cmd_parser.cc:158-160)
  if (a.stepsize == 0.01) a.stepsize = a.stepsize2;
  // n_threads describes CPU threads: accepted and ignored.  n_datacopy > 1 asks for replicas
  // of the whole graph (one per GPU, dw_multi.h); --gpus N for variable-block shards.
  if (a.n_datacopy == 0) a.n_datacopy = 1;
  if (a.n_threads == 0) a.n_threads = (uint64_t)std::max(1L, sysconf(_SC_NPROCESSORS_CONF));
  a.error_text = err.str();
  return a;
}

// The stdout banner of `dw gibbs` (a drop-in keeps the format scripts grep for: "# label : value"
// lines between two rules, src/cmd_parser.cc:272-290), driven by a table: label, value printer.
std::ostream &operator<<(std::ostream &stream, const CmdLine &args) {
  struct Row { const char *label; void (*print)(std::ostream &, const CmdLine &); };
  // (file lists print as "[a, b]", nothing when empty)
#define DWX_ROW_LIST(label, field) {label, [](std::ostream &o, const CmdLine &a) { \
    for (size_t i = 0; i < a.field.size(); ++i) o << (i ? ", " : "[") << a.field[i]; \
    if (!a.field.empty()) o << ']'; }}
#define DWX_ROW(label, field) {label, [](std::ostream &o, const CmdLine &a) { o << a.field; }}
  static const Row rows[] = {
      DWX_ROW("fg_file", fg_file),
      DWX_ROW_LIST("variable_file", variable_file),
      DWX_ROW_LIST("domain_file", domain_file),
      DWX_ROW_LIST("weight_file", weight_file),
      DWX_ROW_LIST("factor_file", factor_file),
      DWX_ROW("output_folder", output_folder),
      DWX_ROW("n_learning_epoch", n_learning_epoch),
      DWX_ROW("n_inference_epoch", n_inference_epoch),
      DWX_ROW("stepsize", stepsize),
      DWX_ROW("decay", decay),
      DWX_ROW("regularization", reg_param),
      DWX_ROW("burn_in", burn_in),
      DWX_ROW("n_datacopy", n_datacopy),
      DWX_ROW("n_threads", n_threads),
      DWX_ROW("learn_non_evidence", should_learn_non_evidence),
      DWX_ROW("is_noise_aware", is_noise_aware),
      // additions of this build
      DWX_ROW("device (HIP)", device),
      DWX_ROW("seed", seed),
  };
#undef DWX_ROW
#undef DWX_ROW_LIST
  const std::string rule_fill(48, '#');
  const std::string title = "GIBBS SAMPLING";
  stream << std::string(17, '#') << title << std::string(17, '#') << std::endl;
  for (const Row &r : rows) {
    std::string label = r.label;
    label.resize(19, ' ');
    stream << "# " << label << ": ";
    r.print(stream, args);
    stream << std::endl;
  }
  stream << rule_fill << std::endl;
  return stream;
}

// ------------------------------------------------------------------ loader
namespace {
// whole-file read-only mapping
struct Mapped {
  const uint8_t *p = nullptr;
  size_t n = 0;
  explicit Mapped(const std::string &path) {
    int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) throw std::runtime_error("cannot open " + path);
    struct stat st;
    if (fstat(fd, &st) != 0) { close(fd); throw std::runtime_error("cannot stat " + path); }
    n = (size_t)st.st_size;
    if (S_ISREG(st.st_mode) && n > 0) {
      void *m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
      if (m == MAP_FAILED) { close(fd); throw std::runtime_error("cannot mmap " + path); }
      p = (const uint8_t *)m;
      mapped_ = true;
    } else {
      // pipes / process substitution (the reference's tests pass <(cat ...)): slurp
      std::vector<uint8_t> *buf = new std::vector<uint8_t>();
      uint8_t tmp[1 << 16];
      ssize_t r;
      while ((r = read(fd, tmp, sizeof tmp)) > 0) buf->insert(buf->end(), tmp, tmp + r);
      owned_ = buf;
      p = buf->data();
      n = buf->size();
    }
    close(fd);
  }
  ~Mapped() {
    if (mapped_) dwx::unmap_parallel((void *)p, n);   // (42 GB of page-table entries at config 5's size)
    delete owned_;
  }
  Mapped(const Mapped &) = delete;

 private:
  bool mapped_ = false;
  std::vector<uint8_t> *owned_ = nullptr;
};

inline uint64_t be64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return be64toh(v); }
inline uint16_t be16(const uint8_t *p) { uint16_t v; memcpy(&v, p, 2); return be16toh(v); }
inline double bef64(const uint8_t *p) { uint64_t v = be64(p); double d; memcpy(&d, &v, 8); return d; }
}  // namespace

// src/binary_format.cc:23-36: numWeights,numVariables,numFactors,numEdges
void read_meta(const std::string &path, LoadedGraph &g) {
  std::ifstream f(path);
  if (!f) throw std::runtime_error("cannot open meta file " + path);
  std::string buf;
  uint64_t *dst[4] = {&g.n_weights, &g.n_variables, &g.n_factors, &g.n_edges};
  for (int i = 0; i < 4; ++i) {
    if (!getline(f, buf, ',')) buf.clear();
    *dst[i] = (uint64_t)atoll(buf.c_str());
  }
}

// src/binary_format.cc:64-126 (27-byte records; stored at index vid)
void load_variables(const std::vector<std::string> &files, LoadedGraph &g) {
  const uint64_t V = g.n_variables;
  g.var_role.assign(V, 0); g.var_init_value.assign(V, 0);
  g.var_dtype.assign(V, 0); g.var_cardinality.assign(V, 0);
  std::vector<uint8_t> seen(V, 0);
  uint64_t count = 0;
  for (const auto &path : files) {
    Mapped m(path);
    if (m.n % 27) throw std::runtime_error(path + ": truncated variable record");
    const uint64_t n_rec = m.n / 27;
    // One file with exactly V records (the usual case): all host threads decode (10^8 records on one thread
    // were 0.8 s of config 5's load).  A variable id met twice then means another one is missing -- the
    // count check below fails as it does on the serial path, whatever the order the records were written in.
    if (files.size() == 1 && n_rec == V && V >= (1u << 16)) {
      const uint32_t T = dwx::host_threads();
      std::vector<uint64_t> n_new(T, 0), n_ev(T, 0);
      dwx::parallel_parts(n_rec, T, [&](uint32_t t, uint64_t b, uint64_t e) {
        uint64_t fresh = 0, ev = 0;
        for (uint64_t i = b; i < e; ++i) {
          const uint8_t *r = m.p + i * 27;
          const uint64_t vid = be64(r);
          if (vid >= V) throw std::runtime_error(path + ": variable id " + std::to_string(vid) + " out of range");
          const uint16_t dt = be16(r + 17);
          if (dt > 1) throw std::runtime_error("[ERROR] Only Boolean and Categorical variables are supported now!");
          g.var_role[vid] = r[8];
          g.var_init_value[vid] = be64(r + 9);
          g.var_dtype[vid] = dt;
          g.var_cardinality[vid] = be64(r + 19);
          if (!__atomic_exchange_n(&seen[vid], (uint8_t)1, __ATOMIC_RELAXED)) { ++fresh; ev += r[8] >= 1; }
        }
        n_new[t] = fresh; n_ev[t] = ev;
      }, 0);
      for (uint32_t t = 0; t < T; ++t) { count += n_new[t]; g.n_evidence += n_ev[t]; g.n_query += n_new[t] - n_ev[t]; }
      continue;
    }
    for (size_t off = 0; off < m.n; off += 27) {
      const uint8_t *r = m.p + off;
      uint64_t vid = be64(r);
      if (vid >= V) throw std::runtime_error(path + ": variable id " + std::to_string(vid) + " out of range");
      uint16_t dt = be16(r + 17);
      if (dt > 1) throw std::runtime_error("[ERROR] Only Boolean and Categorical variables are supported now!");
      g.var_role[vid] = r[8];
      g.var_init_value[vid] = be64(r + 9);
      g.var_dtype[vid] = dt;
      g.var_cardinality[vid] = be64(r + 19);
      if (!seen[vid]) { seen[vid] = 1; ++count; (r[8] >= 1 ? g.n_evidence : g.n_query)++; }
    }
  }
  // FactorGraph::safety_check (src/factor_graph.cc:201-219)
  if (count != V) throw std::runtime_error("variable count " + std::to_string(count) + " != meta " + std::to_string(V));
}

// src/binary_format.cc:48-62 (17-byte records)
void load_weights(const std::vector<std::string> &files, LoadedGraph &g) {
  const uint64_t W = g.n_weights;
  g.w_initial_value.assign(W, 0.0); g.w_is_fixed.assign(W, 0);
  std::vector<uint8_t> seen(W, 0);
  uint64_t count = 0;
  for (const auto &path : files) {
    Mapped m(path);
    if (m.n % 17) throw std::runtime_error(path + ": truncated weight record");
    for (size_t off = 0; off < m.n; off += 17) {
      const uint8_t *r = m.p + off;
      uint64_t wid = be64(r);
      if (wid >= W) throw std::runtime_error(path + ": weight id " + std::to_string(wid) + " out of range");
      g.w_is_fixed[wid] = r[8];
      g.w_initial_value[wid] = bef64(r + 9);
      if (!seen[wid]) { seen[wid] = 1; ++count; }
    }
  }
  if (count != W) throw std::runtime_error("weight count " + std::to_string(count) + " != meta " + std::to_string(W));
}

// src/binary_format.cc:192-226
void load_domains(const std::vector<std::string> &files, LoadedGraph &g) {
  g.dom_vid.clear(); g.dom_offset.assign(1, 0); g.dom_value.clear(); g.dom_truthiness.clear();
  for (const auto &path : files) {
    Mapped m(path);
    size_t off = 0;
    while (off < m.n) {
      if (off + 16 > m.n) throw std::runtime_error(path + ": truncated domain block");
      uint64_t vid = be64(m.p + off), card = be64(m.p + off + 8);
      off += 16;
      if (card > (m.n - off) / 16) throw std::runtime_error(path + ": truncated domain block");
      g.dom_vid.push_back(vid);
      for (uint64_t i = 0; i < card; ++i, off += 16) {
        g.dom_value.push_back(be64(m.p + off));
        g.dom_truthiness.push_back(bef64(m.p + off + 8));
      }
      g.dom_offset.push_back(g.dom_value.size());
    }
  }
}

// src/binary_format.cc:128-190; factor ids are assigned in read order.
// Records have variable length, so a file is first cut at record boundaries -- for free
// when every record has the arity of the first one (what text2bin writes: one arity per
// file), which the parse itself verifies; otherwise by a hop over the arity fields -- and
// the pieces are decoded in parallel straight into the columns.
namespace {
struct Piece { size_t off; uint64_t n_rec, f0, e0; size_t end; };

// decode one piece; returns false if its records do not end where the cut said
bool parse_piece(const Mapped &m, const std::string &path, const Piece &pc, LoadedGraph &g,
                 bool speculative) {
  size_t off = pc.off;
  uint64_t f = pc.f0, e = pc.e0;
  for (uint64_t r = 0; r < pc.n_rec; ++r, ++f) {
    if (off + 10 > m.n) {
      if (speculative) return false;
      throw std::runtime_error(path + ": truncated factor record");
    }
    const uint16_t func = be16(m.p + off);
    const uint64_t arity = be64(m.p + off + 2);
    off += 10;
    if (arity > (m.n - off) / 16 || off + 16 * arity + 16 > m.n) {
      if (speculative) return false;
      throw std::runtime_error(path + ": truncated factor record");
    }
    // more than the meta file announced (e may already be past n_edges in a speculative piece:
    // test it before the subtraction, which would wrap)
    if (f >= g.n_factors || e > g.n_edges || arity > g.n_edges - e) return false;
    g.fac_func[f] = func;
    g.fac_edge_offset[f] = e;
    for (uint64_t i = 0; i < arity; ++i, off += 16, ++e) {
      g.edge_vid[e] = be64(m.p + off);
      g.edge_equal_to[e] = be64(m.p + off + 8);
    }
    g.fac_weight_id[f] = be64(m.p + off);
    g.fac_feature_value[f] = bef64(m.p + off + 8);
    off += 16;
  }
  return off == pc.end;
}
}  // namespace

void load_factors(const std::vector<std::string> &files, LoadedGraph &g) {
  g.fac_func.reset(g.n_factors); g.fac_edge_offset.reset(g.n_factors + 1);
  g.fac_weight_id.reset(g.n_factors); g.fac_feature_value.reset(g.n_factors);
  g.edge_vid.reset(g.n_edges); g.edge_equal_to.reset(g.n_edges);
  const uint32_t nth = dwx::host_threads();
  const uint64_t kPiece = 1 << 16;   // records per piece
  uint64_t f_base = 0, e_base = 0;
  for (const auto &path : files) {
    Mapped m(path);
    if (m.n == 0) continue;
    if (m.n < 10) throw std::runtime_error(path + ": truncated factor record");
    std::vector<Piece> pieces;
    bool done = false;
    // (1) optimistic: fixed stride of the first record
    const uint64_t a0 = be64(m.p + 2);
    if (a0 <= (m.n - 10) / 16) {
      const size_t rs = 26 + 16 * a0;
      if (m.n % rs == 0) {
        const uint64_t n = m.n / rs;
        for (uint64_t r = 0; r < n; r += kPiece) {
          const uint64_t k = std::min(kPiece, n - r);
          pieces.push_back({(size_t)(r * rs), k, f_base + r, e_base + r * a0, (size_t)((r + k) * rs)});
        }
        // the pieces write fac_*[f_base + r] and edge_*[e_base + r * a0 ...]: reject before any
        // thread starts if the meta file announced fewer factors or edges than this file holds
        // (the general path below then reports which count is off)
        std::atomic<bool> ok{f_base <= g.n_factors && n <= g.n_factors - f_base && e_base <= g.n_edges &&
                             (a0 == 0 || n <= (g.n_edges - e_base) / a0)};
        if (ok)
          dwx::parallel_ranges(pieces.size(), nth, [&](uint64_t b, uint64_t e) {
            for (uint64_t i = b; i < e && ok; ++i)
              if (!parse_piece(m, path, pieces[i], g, true)) ok = false;
          }, 2);
        if (ok) { f_base += n; e_base += n * a0; done = true; }
      }
    }
    if (!done) {
      // (2) general: hop from record to record reading only the arity, cut every kPiece
      pieces.clear();
      size_t off = 0;
      uint64_t f = f_base, e = e_base;
      Piece cur{0, 0, f, e, 0};
      while (off < m.n) {
        if (off + 10 > m.n) throw std::runtime_error(path + ": truncated factor record");
        const uint64_t arity = be64(m.p + off + 2);
        if (arity > (m.n - off - 10) / 16 || off + 26 + 16 * arity > m.n)
          throw std::runtime_error(path + ": truncated factor record");
        off += 26 + 16 * arity;
        ++f; e += arity;
        if (++cur.n_rec == kPiece) { cur.end = off; pieces.push_back(cur); cur = Piece{off, 0, f, e, 0}; }
      }
      if (cur.n_rec) { cur.end = off; pieces.push_back(cur); }
      if (f > g.n_factors)
        throw std::runtime_error("factor count " + std::to_string(f) + " != meta " + std::to_string(g.n_factors));
      if (e > g.n_edges)
        throw std::runtime_error("edge count " + std::to_string(e) + " != meta " + std::to_string(g.n_edges));
      dwx::parallel_ranges(pieces.size(), nth, [&](uint64_t b, uint64_t pe) {
        for (uint64_t i = b; i < pe; ++i)
          if (!parse_piece(m, path, pieces[i], g, false)) throw std::runtime_error(path + ": inconsistent factor records");
      }, 2);
      f_base = f; e_base = e;
    }
  }
  if (f_base != g.n_factors)
    throw std::runtime_error("factor count " + std::to_string(f_base) + " != meta " + std::to_string(g.n_factors));
  if (e_base != g.n_edges)
    throw std::runtime_error("edge count " + std::to_string(e_base) + " != meta " + std::to_string(g.n_edges));
  g.fac_edge_offset[g.n_factors] = g.n_edges;
}

// ------------------------------------------------------------------ sharded loader
// `dw gibbs --gpus N` (dw_multi.cc): the factor files are decoded ONCE, straight into the
// ranks' shard graphs -- no whole-graph factor columns in between (config 5's 10^9 factors:
// 42 GB of file become 8 shards of a ninth each instead of 60 GB of columns + the shards).
// Same reading of the format as load_factors (src/binary_format.cc:128-190: factor ids in
// read order, so a shard's factors keep their relative order), same cut into pieces, same
// errors.  Pass 1 walks every piece, counts per (piece, rank) the factors that touch the
// rank's block and their edges, and collects the ranks' ghosts; pass 2 walks the pieces again
// and copies, ids renumbered (owned: id - begin, ghosts behind them in ascending global id).
namespace {
struct Cut { std::vector<Piece> pieces; uint64_t n_rec = 0, n_edge = 0; };

// every record has the arity of the first one (to be verified by the walk)?
bool cut_fixed(const Mapped &m, uint64_t f_base, uint64_t e_base, Cut &c) {
  const uint64_t kPiece = 1 << 16;
  c = Cut();
  const uint64_t a0 = be64(m.p + 2);
  if (a0 > (m.n - 10) / 16) return false;
  const size_t rs = 26 + 16 * a0;
  if (m.n % rs != 0) return false;
  const uint64_t n = m.n / rs;
  for (uint64_t r = 0; r < n; r += kPiece) {
    const uint64_t k = std::min(kPiece, n - r);
    c.pieces.push_back({(size_t)(r * rs), k, f_base + r, e_base + r * a0, (size_t)((r + k) * rs)});
  }
  c.n_rec = n; c.n_edge = n * a0;
  return true;
}

// hop from record to record reading only the arity
void cut_hop(const Mapped &m, const std::string &path, uint64_t f_base, uint64_t e_base, Cut &c) {
  const uint64_t kPiece = 1 << 16;
  c = Cut();
  size_t off = 0;
  uint64_t f = f_base, e = e_base;
  Piece cur{0, 0, f, e, 0};
  while (off < m.n) {
    if (off + 10 > m.n) throw std::runtime_error(path + ": truncated factor record");
    const uint64_t arity = be64(m.p + off + 2);
    if (arity > (m.n - off - 10) / 16 || off + 26 + 16 * arity > m.n)
      throw std::runtime_error(path + ": truncated factor record");
    off += 26 + 16 * arity;
    ++f; e += arity;
    if (++cur.n_rec == kPiece) { cur.end = off; c.pieces.push_back(cur); cur = Piece{off, 0, f, e, 0}; }
  }
  if (cur.n_rec) { cur.end = off; c.pieces.push_back(cur); }
  c.n_rec = f - f_base; c.n_edge = e - e_base;
}

// fn(func, arity, pointer to the arity 16-byte entries (the weight id and the feature value
// follow them)) for every record of a piece; false if the records do not end where the cut said
template <class Fn>
bool walk_piece(const Mapped &m, const Piece &pc, Fn &&fn) {
  size_t off = pc.off;
  for (uint64_t r = 0; r < pc.n_rec; ++r) {
    if (off + 10 > m.n) return false;
    const uint16_t func = be16(m.p + off);
    const uint64_t arity = be64(m.p + off + 2);
    off += 10;
    if (arity > (m.n - off) / 16 || off + 16 * arity + 16 > m.n) return false;
    fn(func, arity, m.p + off);
    off += 16 * arity + 16;
  }
  return off == pc.end;
}
}  // namespace

void load_factors_sharded(const std::vector<std::string> &files, const LoadedGraph &whole, int world,
                          std::vector<ShardGraph> &shards) {
  if (world < 1 || world > 64) throw std::runtime_error("load_factors_sharded: 1 to 64 ranks");
  const uint32_t nth = dwx::host_threads();
  const uint64_t V = whole.n_variables, per = std::max<uint64_t>(1, (V + world - 1) / world);
  shards.clear();
  shards.resize(world);
  for (int r = 0; r < world; ++r) shard_range(V, r, world, shards[r].begin, shards[r].end);
  // (the rank that owns v; rank and mask are only used for v < V)
  auto owner = [&](uint64_t v) { return (int)(v / per); };

  struct File { std::unique_ptr<Mapped> m; std::string path; Cut cut; size_t piece0; };
  std::vector<File> fs;
  std::vector<uint64_t> cnt_f, cnt_e;              // [piece * world + rank]
  std::vector<std::vector<uint64_t>> gh(world);    // ghosts per rank, ascending, unique
  auto tidy = [](std::vector<uint64_t> &v) {
    std::sort(v.begin(), v.end());
    v.erase(std::unique(v.begin(), v.end()), v.end());
  };
  uint64_t f_base = 0, e_base = 0;
  size_t n_pieces = 0;
  for (const auto &path : files) {
    File fl;
    fl.m.reset(new Mapped(path));
    fl.path = path;
    const Mapped &m = *fl.m;
    if (m.n == 0) continue;
    if (m.n < 10) throw std::runtime_error(path + ": truncated factor record");
    // pass 1 of this file: counts and ghosts; first on the optimistic cut, then on the hop
    for (int attempt = 0; attempt < 2; ++attempt) {
      if (attempt == 0) {
        if (!cut_fixed(m, f_base, e_base, fl.cut)) continue;
      } else {
        cut_hop(m, path, f_base, e_base, fl.cut);
      }
      const Cut &c = fl.cut;
      // more than the meta file announced: the general path reports which count is off
      if (f_base > whole.n_factors || c.n_rec > whole.n_factors - f_base)
        { if (attempt == 0) continue;
          throw std::runtime_error("factor count " + std::to_string(f_base + c.n_rec) + " != meta " + std::to_string(whole.n_factors)); }
      if (e_base > whole.n_edges || c.n_edge > whole.n_edges - e_base)
        { if (attempt == 0) continue;
          throw std::runtime_error("edge count " + std::to_string(e_base + c.n_edge) + " != meta " + std::to_string(whole.n_edges)); }
      const size_t np = c.pieces.size();
      std::vector<uint64_t> cf(np * world, 0), ce(np * world, 0);
      const uint32_t T = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(nth, np));
      std::vector<std::vector<std::vector<uint64_t>>> part(T, std::vector<std::vector<uint64_t>>(world));
      std::atomic<bool> ok{true};
      std::atomic<uint64_t> bad_vid{~0ull};
      dwx::parallel_parts(np, T, [&](uint32_t t, uint64_t pb, uint64_t pe) {
        std::vector<size_t> sorted_upto(world, 0);
        for (uint64_t i = pb; i < pe && ok; ++i) {
          uint64_t *nf = &cf[i * world], *ne = &ce[i * world];
          const bool good = walk_piece(m, c.pieces[i], [&](uint16_t, uint64_t arity, const uint8_t *ent) {
            uint64_t mask = 0;
            for (uint64_t k = 0; k < arity; ++k) {
              const uint64_t v = be64(ent + 16 * k);
              if (v >= V) { bad_vid = v; return; }
              mask |= 1ull << owner(v);
            }
            for (uint64_t mm = mask; mm; mm &= mm - 1) {
              const int r = __builtin_ctzll(mm);
              ++nf[r]; ne[r] += arity;
              if (mask == (1ull << r)) continue;      // nothing remote
              for (uint64_t k = 0; k < arity; ++k) {
                const uint64_t v = be64(ent + 16 * k);
                if (owner(v) != r) part[t][r].push_back(v);
              }
            }
          });
          if (!good) ok = false;
          for (int r = 0; r < world; ++r)             // (a ghost is named by every factor that reads it)
            if (part[t][r].size() - sorted_upto[r] > (1u << 20)) { tidy(part[t][r]); sorted_upto[r] = part[t][r].size(); }
        }
        for (int r = 0; r < world; ++r) tidy(part[t][r]);
      }, 2);
      if (!ok) {
        if (attempt == 0) continue;
        throw std::runtime_error(path + ": inconsistent factor records");
      }
      if (bad_vid.load() != ~0ull) throw std::runtime_error("factor references unknown variable");
      cnt_f.insert(cnt_f.end(), cf.begin(), cf.end());
      cnt_e.insert(cnt_e.end(), ce.begin(), ce.end());
      for (uint32_t t = 0; t < T; ++t)
        for (int r = 0; r < world; ++r) gh[r].insert(gh[r].end(), part[t][r].begin(), part[t][r].end());
      for (int r = 0; r < world; ++r) tidy(gh[r]);
      fl.piece0 = n_pieces;
      n_pieces += np;
      f_base += c.n_rec; e_base += c.n_edge;
      fs.push_back(std::move(fl));
      break;
    }
  }
  if (f_base != whole.n_factors)
    throw std::runtime_error("factor count " + std::to_string(f_base) + " != meta " + std::to_string(whole.n_factors));
  if (e_base != whole.n_edges)
    throw std::runtime_error("edge count " + std::to_string(e_base) + " != meta " + std::to_string(whole.n_edges));

  // exclusive prefix over the pieces, per rank
  std::vector<uint64_t> off_f((n_pieces + 1) * world, 0), off_e((n_pieces + 1) * world, 0);
  for (size_t i = 0; i < n_pieces; ++i)
    for (int r = 0; r < world; ++r) {
      off_f[(i + 1) * world + r] = off_f[i * world + r] + cnt_f[i * world + r];
      off_e[(i + 1) * world + r] = off_e[i * world + r] + cnt_e[i * world + r];
    }
  for (int r = 0; r < world; ++r) {
    ShardGraph &sg = shards[r];
    sg.ghosts.swap(gh[r]);
    sg.n_ghost = sg.ghosts.size();
    fill_shard_variables(whole, sg);
    LoadedGraph &g = sg.g;
    const uint64_t nf = off_f[n_pieces * world + r], ne = off_e[n_pieces * world + r];
    g.n_factors = nf; g.n_edges = ne;
    g.fac_func.reset(nf); g.fac_edge_offset.reset(nf + 1); g.fac_weight_id.reset(nf); g.fac_feature_value.reset(nf);
    g.edge_vid.reset(ne); g.edge_equal_to.reset(ne);
    g.fac_edge_offset[nf] = ne;
  }
  // pass 2: copy
  for (const File &fl : fs) {
    const Mapped &m = *fl.m;
    dwx::parallel_ranges(fl.cut.pieces.size(), nth, [&](uint64_t pb, uint64_t pe) {
      std::vector<uint64_t> nf(world), ne(world);
      for (uint64_t i = pb; i < pe; ++i) {
        const size_t gi = fl.piece0 + i;
        for (int r = 0; r < world; ++r) { nf[r] = off_f[gi * world + r]; ne[r] = off_e[gi * world + r]; }
        walk_piece(m, fl.cut.pieces[i], [&](uint16_t func, uint64_t arity, const uint8_t *ent) {
          uint64_t mask = 0;
          for (uint64_t k = 0; k < arity; ++k) mask |= 1ull << owner(be64(ent + 16 * k));
          const uint64_t wid = be64(ent + 16 * arity);
          const double fval = bef64(ent + 16 * arity + 8);
          for (uint64_t mm = mask; mm; mm &= mm - 1) {
            const int r = __builtin_ctzll(mm);
            ShardGraph &sg = shards[r];
            LoadedGraph &g = sg.g;
            const uint64_t f = nf[r]++;
            uint64_t o = ne[r];
            g.fac_func[f] = func;
            g.fac_edge_offset[f] = o;
            g.fac_weight_id[f] = wid;
            g.fac_feature_value[f] = fval;
            for (uint64_t k = 0; k < arity; ++k, ++o) {
              const uint64_t v = be64(ent + 16 * k);
              g.edge_vid[o] = owner(v) == r ? v - sg.begin
                  : (sg.end - sg.begin) + (uint64_t)(std::lower_bound(sg.ghosts.begin(), sg.ghosts.end(), v) - sg.ghosts.begin());
              g.edge_equal_to[o] = be64(ent + 16 * k + 8);
            }
            ne[r] = o;
          }
        });
      }
    }, 2);
  }
}

dwx_graph_desc LoadedGraph::desc() const {
  dwx_graph_desc d;
  memset(&d, 0, sizeof d);
  d.num_variables = n_variables; d.num_factors = n_factors; d.num_edges = n_edges; d.num_weights = n_weights;
  d.var_role = var_role.data(); d.var_init_value = var_init_value.data();
  d.var_dtype = var_dtype.data(); d.var_cardinality = var_cardinality.data();
  d.num_domains = dom_vid.size();
  d.dom_vid = dom_vid.data(); d.dom_offset = dom_offset.data();
  d.dom_value = dom_value.data(); d.dom_truthiness = dom_truthiness.data();
  d.fac_func = fac_func.data(); d.fac_edge_offset = fac_edge_offset.data();
  d.fac_weight_id = fac_weight_id.data(); d.fac_feature_value = fac_feature_value.data();
  d.edge_vid = edge_vid.data(); d.edge_equal_to = edge_equal_to.data();
  d.w_initial_value = w_initial_value.data(); d.w_is_fixed = w_is_fixed.data();
  return d;
}

// ------------------------------------------------------------------ dumps
// Both dumps format variable ranges in parallel into per-thread buffers and write them
// out in order.  "%g" is what `ostream << double` prints at the default precision 6, so the
// bytes equal the reference's (tests/test_dw_cli.py, tests/test_end_to_end.py).
namespace {
template <class LineFn>
void dump_parallel(std::ostream &o, uint64_t n, LineFn &&line) {
  const uint32_t nth = dwx::host_threads();
  std::vector<std::string> bufs(nth);
  dwx::parallel_parts(n, nth, [&](uint32_t t, uint64_t b, uint64_t e) {
    std::string &s = bufs[t];
    s.reserve((e - b) * 24);
    for (uint64_t i = b; i < e; ++i) line(i, s);
  });
  for (const std::string &s : bufs) o.write(s.data(), (std::streamsize)s.size());
}
// The same straight into a file descriptor: the parts are formatted by all threads and written in order by
// plain write()s of whole buffers.  (Copying them into a shared mapping of the file from all threads was
// measured and dropped: on the box's overlay file system the mapping's page faults made 1.1 GB of marginals
// take 1.1 s against 0.4-0.6 s for the writes.)  The bytes are those of the stream version.
template <class LineFn>
void dump_parallel_to_file(const std::string &path, uint64_t n, LineFn &&line) {
  const int fd = open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
  if (fd < 0) throw std::runtime_error("cannot write " + path);
  struct Closer { int fd; ~Closer() { close(fd); } } closer{fd};
  const uint32_t nth = dwx::host_threads();
  std::vector<std::string> bufs(nth);
  dwx::parallel_parts(n, nth, [&](uint32_t t, uint64_t b, uint64_t e) {
    std::string &s = bufs[t];
    s.reserve((e - b) * 24);
    for (uint64_t i = b; i < e; ++i) line(i, s);
  });
  for (const std::string &s : bufs)
    for (size_t done = 0; done < s.size();) {
      const ssize_t w = write(fd, s.data() + done, s.size() - done);
      if (w <= 0) throw std::runtime_error("cannot write " + path);
      done += (size_t)w;
    }
}
// A marginal is tally / nsamples with nsamples the run's -i for every sampled variable: at most -i + 1
// distinct values.  Their "%g" strings are made once, by snprintf itself (the bytes stay the reference's),
// and a line is two integers written by hand plus a lookup: 50 M lines of config 5 cost 15 CPU-seconds
// through snprintf -- a second of wall time under a 16-CPU quota.
struct RatioStrings {
  uint64_t n = 0;
  std::vector<std::string> str;   // [t] = "%g" of 1.0 * t / n, t in [0, n]
  void build(uint64_t nsamples) {
    n = 0; str.clear();
    if (nsamples == 0 || nsamples > 65536) return;
    n = nsamples;
    str.resize(n + 1);
    char tmp[64];
    for (uint64_t t = 0; t <= n; ++t) str[t].assign(tmp, (size_t)snprintf(tmp, sizeof tmp, "%g", 1.0 * t / n));
  }
};
inline void append_u64(std::string &s, uint64_t v) {
  char tmp[24];
  int i = 24;
  do { tmp[--i] = (char)('0' + v % 10); v /= 10; } while (v);
  s.append(tmp + i, (size_t)(24 - i));
}
inline void append_line(std::string &s, uint64_t a, double x) {
  char tmp[64];
  s.append(tmp, (size_t)snprintf(tmp, sizeof tmp, "%llu %g\n", (unsigned long long)a, x));
}
inline void append_line(std::string &s, uint64_t a, uint64_t b, double x) {
  char tmp[96];
  s.append(tmp, (size_t)snprintf(tmp, sizeof tmp, "%llu %llu %g\n", (unsigned long long)a, (unsigned long long)b, x));
}
// "<id> <value> <tally / nsamples>\n"
inline void append_marginal(std::string &s, const RatioStrings &rs, uint64_t id, uint64_t value, uint64_t tally, uint64_t nsamples) {
  if (nsamples == rs.n && rs.n && tally <= rs.n) {
    append_u64(s, id); s.push_back(' '); append_u64(s, value); s.push_back(' ');
    s.append(rs.str[tally]); s.push_back('\n');
  } else {
    append_line(s, id, value, 1.0 * tally / nsamples);
  }
}
// (the -i of the run: the count of the first sampled variable)
inline uint64_t common_nsamples(const LoadedGraph &g, bool sample_evidence, const uint64_t *nsamples, uint64_t n_vars) {
  for (uint64_t v = 0; v < n_vars; ++v)
    if (g.var_role[v] < 1 || sample_evidence) return nsamples[v];
  return 0;
}
}  // namespace

// src/inference_result.cc:101-105
void dump_weights_in_text(std::ostream &o, const std::vector<double> &w) {
  dump_parallel(o, w.size(), [&](uint64_t j, std::string &s) { append_line(s, j, w[j]); });
}

void dump_weights_to_file(const std::string &path, const std::vector<double> &w) {
  dump_parallel_to_file(path, w.size(), [&](uint64_t j, std::string &s) { append_line(s, j, w[j]); });
}

void dump_marginals_to_file(const std::string &path, const LoadedGraph &g, bool sample_evidence,
                            const uint64_t *var_val_base, const uint64_t *value_sparse,
                            const uint64_t *tallies, const uint64_t *nsamples) {
  RatioStrings rs;
  rs.build(common_nsamples(g, sample_evidence, nsamples, g.n_variables));
  dump_parallel_to_file(path, g.n_variables, [&](uint64_t v, std::string &s) {
    if (g.var_role[v] >= 1 && !sample_evidence) return;
    const uint64_t b = var_val_base[v];
    if (g.var_dtype[v] == 0) {
      append_marginal(s, rs, v, (uint64_t)1, tallies[b], nsamples[v]);
    } else {
      for (uint64_t j = 0; j < g.var_cardinality[v]; ++j)
        append_marginal(s, rs, v, value_sparse[b + j], tallies[b + j], nsamples[v]);
    }
  });
}

// src/inference_result.cc:211-243
void dump_marginals_in_text(std::ostream &o, const LoadedGraph &g, bool sample_evidence,
                            const uint64_t *var_val_base, const uint64_t *value_sparse,
                            const uint64_t *tallies, const uint64_t *nsamples, uint64_t id_offset, uint64_t n_vars) {
  const uint64_t nv = std::min<uint64_t>(g.n_variables, n_vars);
  RatioStrings rs;
  rs.build(common_nsamples(g, sample_evidence, nsamples, nv));
  dump_parallel(o, nv, [&](uint64_t v, std::string &s) {
    if (g.var_role[v] >= 1 && !sample_evidence) return;
    const uint64_t b = var_val_base[v];
    if (g.var_dtype[v] == 0) {
      append_marginal(s, rs, v + id_offset, (uint64_t)1, tallies[b], nsamples[v]);
    } else {
      for (uint64_t j = 0; j < g.var_cardinality[v]; ++j)
        append_marginal(s, rs, v + id_offset, value_sparse[b + j], tallies[b + j], nsamples[v]);
    }
  });
}

// ------------------------------------------------------------------ driver
namespace {
struct Check {
  void operator()(int rc) const {
    if (rc != DWX_OK) throw std::runtime_error(std::string("dwx: ") + dwx_last_error());
  }
};
double now() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
}  // namespace

// Graph-compile options of a `dw gibbs` run.  Ordering the variables of an all-unary graph by the
// weight of their first record makes a sweep ~5 % faster and the host-side build
// much slower -- the records are then gathered in a random order: 10 s of a 42 s run at config 5's
// size (W = 10 M), 0.5 s of 3.1 s at config 3's, for 0.01 s of sweeps saved.  It pays back after
// ~36 000 sweeps; a run announces its sweeps (-l + -i), so only long runs get it.  Results do not
// depend on the order (Philox counters are global ids, sums are integers).  DWX_DW_WEIGHT_ORDER=1 / 0
// forces it on / off.
dwx_compile_opts compile_opts_for(const CmdLine &args) {
  dwx_compile_opts co;
  std::memset(&co, 0, sizeof co);
  bool order = args.n_learning_epoch + args.n_inference_epoch >= 20000;
  if (const char *e = getenv("DWX_DW_WEIGHT_ORDER")) order = atoi(e) != 0;
  co.no_weight_order = order ? 0u : 1u;
  // (what the gradient all-reduce may assume about the sums: only runs over several GPUs ask)
  co.no_narrow_info = (args.gpus >= 2 || args.n_datacopy > 1) ? 0u : 1u;
  return co;
}

// A `dw gibbs` run that has written its result files is DONE: unmapping tens of gigabytes of host
// columns and freeing the device buffers one by one (2.5 s at config 5's size, a tenth of the run)
// is work the process exit does at once.  Everything buffered is flushed first; the exit code is
// the run's.  DWX_FULL_TEARDOWN=1 (leak checks) keeps the orderly teardown.
void quick_exit_if_done(int exit_code) {
  if (getenv("DWX_FULL_TEARDOWN")) return;
  std::cout.flush();
  std::cerr.flush();
  fflush(nullptr);
  // the pages of every large host array, handed back by all threads (the exit would clear them on one core;
  // this only pays since the uploads are staged -- rt_hip.h -- and the arrays' pages therefore ordinary ones)
  const auto t_drop = std::chrono::steady_clock::now();
  if (!getenv("DWX_NO_EXIT_DROP")) dwx::mapping_registry_drop_all();   // (A/B knob)
  if (getenv("DWX_TIMING"))
    fprintf(stderr, "[dw timing] host arrays handed back: %g s\n",
            std::chrono::duration<double>(std::chrono::steady_clock::now() - t_drop).count());
  if (getenv("DWX_TIMING")) {
    fprintf(stderr, "[dw timing] teardown (device + graph): skipped (process exit)\n");
    // (tools/e2e_walltime.py: what the process exit itself costs = the caller's clock - this)
    fprintf(stderr, "[dw timing] epoch at exit: %.3f\n",
            std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count());
  }
  std::_Exit(exit_code);
}

int gibbs(const CmdLine &args) {
  Check ok;
  dwx_graph *graph = nullptr;
  dwx_sampler *sampler = nullptr;
  int exit_code = 0;
  try {
    if (!args.should_be_quiet) {
      std::cout << std::endl;
      std::cout << "#################MACHINE CONFIG#################" << std::endl;
      std::cout << "# # HIP device       : " << args.device << std::endl;
      std::cout << "################################################" << std::endl;
      std::cout << std::endl;
      std::cout << args << std::endl;
    }
    // DWX_TIMING=1: wall time of every host phase on stderr
    const bool timing = getenv("DWX_TIMING") != nullptr;
    if (timing)
      fprintf(stderr, "[dw timing] epoch at start: %.3f\n",
              std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count());
    double t_phase = now();
    auto phase = [&](const char *what) {
      const double t = now();
      if (timing) std::cerr << "[dw timing] " << what << ": " << t - t_phase << " s" << std::endl;
      t_phase = t;
    };
    // the HIP runtime starts up (0.2 s) on a helper thread while the files are read and the
    // graph is compiled; a failure there is reported by dwx_sampler_create
    std::thread hip_start([&]() { (void)dwx_device_init(args.device); });
    struct Joiner { std::thread &t; ~Joiner() { if (t.joinable()) t.join(); } } hip_start_joiner{hip_start};
    LoadedGraph lg;
    read_meta(args.fg_file, lg);
    std::cout << "Factor graph to load:\t#V=" << lg.n_variables << " #F=" << lg.n_factors
              << " #W=" << lg.n_weights << " #E=" << lg.n_edges << " #Val=0" << std::endl;
    std::cout << "\tinitializing factor graph..." << std::endl;
    std::cout << "\tloading factor graph..." << std::endl;
    // load order of the reference (src/dimmwitted.cc:66-69)
    load_variables(args.variable_file, lg);
    load_weights(args.weight_file, lg);
    load_domains(args.domain_file, lg);
    phase("load variables/weights/domains");
    load_factors(args.factor_file, lg);
    phase("load factors");
    dwx_graph_desc desc = lg.desc();
    dwx_compile_opts co = compile_opts_for(args);
    ok(dwx_graph_create(&desc, &co, &graph));
    phase("dwx_graph_create (index, colouring, device layout)");
    // the decoded factor / edge columns (42 GB at config 5's size) are dead once the graph is compiled (it keeps
    // no pointer into them): a helper thread gives them back while the sampler is created -- in 64 MiB steps
    // that take the address-space lock shared (host_parallel.h, unmap_parallel), so the creating thread's own
    // mappings never wait longer than one step
    {
      struct Dead {
        dwx::RawArray<uint16_t> a; dwx::RawArray<uint64_t> b, c; dwx::RawArray<double> d; dwx::RawArray<uint64_t> e, f;
      };
      Dead *dead = new Dead{std::move(lg.fac_func), std::move(lg.fac_edge_offset), std::move(lg.fac_weight_id),
                            std::move(lg.fac_feature_value), std::move(lg.edge_vid), std::move(lg.edge_equal_to)};
      std::thread([dead]() { delete dead; }).detach();
    }
    dwx_graph_info info;
    ok(dwx_graph_get_info(graph, &info));
    auto print_size = [&](const char *what) {
      std::cout << what << "#V=" << lg.n_variables << "(#Vqry=" << lg.n_query << " #Vevd=" << lg.n_evidence
                << ") #F=" << lg.n_factors << " #W=" << lg.n_weights << " #E=" << lg.n_edges
                << " #Val=" << info.num_values << std::endl;
    };
    print_size("Factor graph loaded:\t");
    print_size("Factor graph indexed:\t");
    if (!args.should_be_quiet)
      std::cout << "Device layout: " << info.num_colors << " colour(s), " << info.num_tiles
                << " tile(s), " << info.device_bytes << " bytes in HBM" << std::endl;

    dwx_options o;
    dwx_default_options(&o);
    o.device = args.device;
    o.sample_evidence = args.should_sample_evidence;
    o.learn_non_evidence = args.should_learn_non_evidence;
    o.noise_aware = args.is_noise_aware;
    o.regularization = args.regularization_l1 ? 0 : 1;
    o.reg_param = args.reg_param;
    o.seed = args.seed;
    o.step_cap = args.step_cap;
    o.plan_layouts = args.plan_layouts;
    hip_start.join();
    ok(dwx_sampler_create(graph, &o, &sampler));
    phase("dwx_sampler_create (upload, gradient incidence list, curvature estimate)");

    const uint64_t V = lg.n_variables, W = lg.n_weights;
    const bool progress = !args.should_be_quiet;
    std::vector<double> weights(W), prev(lg.w_initial_value);

    // ---- DimmWitted::learn (src/dimmwitted.cc:162-207), one sampler
    double t_total = now(), stepsize = args.stepsize;
    for (uint64_t e = 0; e < args.n_learning_epoch; ++e) {
      if (progress) std::cout << std::setprecision(3) << "LEARNING EPOCH " << e << "~" << e << "...." << std::flush;
      double t0 = now();
      // the plan this sweep will run with (cached in the library; dwx_sample_sgd_async makes the
      // same one): reported below whenever it differs from "one batch, the step as given"
      uint32_t plan_batches = 1, plan_chunks = 1;
      double plan_min_step = stepsize;
      if (progress) ok(dwx_sgd_plan(sampler, stepsize, 0, &plan_batches, &plan_chunks, &plan_min_step));
      ok(dwx_sample_sgd_async(sampler, stepsize));
      ok(dwx_wait(sampler));
      double elapsed = now() - t0;
      if (progress) {
        // update_weights' norms (src/dimmwitted.cc:218-236)
        ok(dwx_get_weights(sampler, weights.data()));
        double lmax = -INFINITY, l2 = 0.0;
        for (uint64_t j = 0; j < W; ++j) {
          double diff = fabs(weights[j] - prev[j]);
          l2 += diff * diff;
          if (lmax < diff) lmax = diff;
        }
        lmax /= stepsize;
        std::cout << std::setprecision(3) << "" << elapsed << " sec." << "," << V / elapsed
                  << " vars/sec." << ",stepsize=" << stepsize << ",lmax=" << lmax
                  << ",l2=" << sqrt(l2) / stepsize;
        // (this build's addition) mini-batches per sweep and the smallest step any weight took:
        // heavily tied weights saturate below the requested step (include/dwx.h, dwx_sgd_plan)
        if (plan_batches > 1 || plan_min_step < 0.999 * stepsize)
          std::cout << ",batches=" << plan_batches << ",min_step=" << plan_min_step;
        std::cout << std::endl << std::setprecision(6);
        prev = weights;
      }
      stepsize *= args.decay;
    }
    std::cout << std::setprecision(6) << "TOTAL LEARNING TIME: " << now() - t_total << " sec." << std::endl;

    phase("learning sweeps");
    // ---- dump_weights (src/dimmwitted.cc:245-258): before inference starts
    ok(dwx_get_weights(sampler, weights.data()));
    if (progress) {
      std::cout << "LEARNING SNIPPETS (QUERY WEIGHTS):" << std::endl;
      for (uint64_t j = 0; j < W && j < 10; ++j) std::cout << "   " << j << " " << weights[j] << std::endl;
      std::cout << "   ..." << std::endl;
    }
    {
      std::string fn = args.output_folder + "/inference_result.out.weights.text";
      std::cout << "DUMPING... TEXT    : " << fn << std::endl;
      dump_weights_to_file(fn, weights);
    }

    phase("dump weights");
    // ---- DimmWitted::inference (src/dimmwitted.cc:121-160)
    t_total = now();
    ok(dwx_clear_tallies(sampler));
    for (uint64_t e = 0; e < args.n_inference_epoch; ++e) {
      if (progress) {
        std::cout << std::setprecision(3) << "INFERENCE EPOCH " << e << "~" << e << "...." << std::flush;
        double t0 = now();
        ok(dwx_sample_async(sampler));
        ok(dwx_wait(sampler));
        double elapsed = now() - t0;
        std::cout << std::setprecision(3) << "" << elapsed << " sec." << "," << V / elapsed << " vars/sec"
                  << std::endl << std::setprecision(6);
      } else {
        // quiet: nothing is printed per epoch -- all of them in one call (an all-unary graph
        // runs them in one launch, any other queues them on the stream)
        ok(dwx_sample_n_async(sampler, (uint32_t)std::min<uint64_t>(args.n_inference_epoch - e, 1u << 20)));
        e += std::min<uint64_t>(args.n_inference_epoch - e, 1u << 20) - 1;
      }
    }
    ok(dwx_wait(sampler));
    std::cout << std::setprecision(6) << "TOTAL INFERENCE TIME: " << now() - t_total << " sec." << std::endl;

    phase("inference sweeps");
    // ---- aggregate_results_and_dump (src/dimmwitted.cc:260-277), only if -i > 0
    if (args.n_inference_epoch > 0) {
      // (uninitialised, first touched by the library's parallel fills: zero-filling 3.2 GB of vectors on one
      // thread was 0.3 s at config 5's size; every entry is written -- one GPU owns every variable)
      dwx::RawArray<uint64_t> tallies(info.num_values), nsamples(V), base(V), sparse(info.num_values);
      ok(dwx_get_tallies(sampler, tallies.data(), nsamples.data()));
      ok(dwx_graph_get_values(graph, base.data(), sparse.data()));
      auto sampled = [&](uint64_t v) { return lg.var_role[v] < 1 || args.should_sample_evidence; };
      if (progress) {
        // InferenceResult::show_marginal_snippet (src/inference_result.cc:129-169): the first
        // ten sampled variables
        std::cout << "INFERENCE SNIPPETS (QUERY VARIABLES):" << std::endl;
        size_t ct = 0;
        for (uint64_t v = 0; v < V; ++v) {
          if (!sampled(v)) continue;
          ++ct;
          std::cout << "   " << v << "  NSAMPLE=" << nsamples[v] << std::endl;
          const uint64_t n = lg.var_dtype[v] == 0 ? 1 : lg.var_cardinality[v];
          for (uint64_t j = 0; j < n; ++j)
            std::cout << "      @ " << (lg.var_dtype[v] == 0 ? 1 : sparse[base[v] + j]) << " -> EXP="
                      << 1.0 * tallies[base[v] + j] / nsamples[v] << std::endl;
          if (ct % 10 == 0) break;
        }
        std::cout << "   ..." << std::endl;
      }
      std::string fn = args.output_folder + "/inference_result.out.text";
      std::cout << "DUMPING... TEXT    : " << fn << std::endl;
      phase("dwx_get_tallies");
      dump_marginals_to_file(fn, lg, args.should_sample_evidence, base.data(), sparse.data(), tallies.data(), nsamples.data());
      phase("dump marginals");
      if (progress) {
        // the reference's closing calibration table (InferenceResult::show_marginal_histogram,
        // src/inference_result.cc:171-209): how many sampled (variable, value) rows have their
        // marginal in each tenth of [0, 1]; a marginal of exactly 1 counts in the last tenth
        constexpr size_t kBins = 10;
        size_t per_bin[kBins] = {0};
        for (uint64_t v = 0; v < V; ++v) {
          if (!sampled(v) || nsamples[v] == 0) continue;
          const uint64_t rows = lg.var_dtype[v] == 0 ? 1 : lg.var_cardinality[v];
          for (uint64_t k = 0; k < rows; ++k) {
            const double p = (double)tallies[base[v] + k] / (double)nsamples[v];
            ++per_bin[std::min(kBins - 1, (size_t)(p * kBins))];
          }
        }
        std::cout << "INFERENCE CALIBRATION (QUERY BINS):" << std::endl;
        char line[96];
        for (size_t i = 0; i < kBins; ++i) {
          snprintf(line, sizeof line, "PROB BIN %.1f~%.1f  -->  # %zu", (double)((float)i / kBins),
                   (double)((float)(i + 1) / kBins), per_bin[i]);
          std::cout << line << std::endl;
        }
      }
    }
    // (before this scope's destructors: they would unmap gigabytes one array after the other)
    quick_exit_if_done(0);
  } catch (const std::exception &e) {
    std::cerr << "dw: " << e.what() << std::endl;
    exit_code = 1;
  }
  const double t_down = now();
  quick_exit_if_done(exit_code);     // (a finished run does not pay for freeing what the process exit frees)
  dwx_sampler_destroy(sampler);
  dwx_graph_destroy(graph);
  if (getenv("DWX_TIMING")) std::cerr << "[dw timing] teardown (device + graph): " << now() - t_down << " s" << std::endl;
  return exit_code;
}

// ------------------------------------------------------------------ text2bin / bin2text
namespace {
void put_be64(std::ostream &o, uint64_t v) { v = htobe64(v); o.write((const char *)&v, 8); }
void put_be16(std::ostream &o, uint16_t v) { v = htobe16(v); o.write((const char *)&v, 2); }
void put_bef64(std::ostream &o, double d) { uint64_t v; memcpy(&v, &d, 8); put_be64(o, v); }

std::vector<std::string> split_tabs(const std::string &line) {
  std::vector<std::string> out;
  size_t b = 0;
  for (;;) {
    size_t e = line.find('\t', b);
    out.push_back(line.substr(b, e == std::string::npos ? std::string::npos : e - b));
    if (e == std::string::npos) break;
    b = e + 1;
  }
  return out;
}

// "{a,b,c}" -> elements (parse_pgarray, src/text2bin.cc:75-97)
std::vector<std::string> parse_pgarray(const std::string &s) {
  if (s.size() < 2 || s.front() != '{' || s.back() != '}')
    throw std::runtime_error("Expected an array '{...}' but found: " + s);
  std::vector<std::string> out;
  std::string body = s.substr(1, s.size() - 2), el;
  std::istringstream ss(body);
  while (getline(ss, el, ',')) out.push_back(el);
  return out;
}
}  // namespace

int text2bin(const CmdLine &a) {
  try {
    std::ifstream fin(a.text2bin_input);
    if (!fin) throw std::runtime_error("cannot open " + a.text2bin_input);
    std::ofstream fout(a.text2bin_output, std::ios::binary);
    if (!fout) throw std::runtime_error("cannot write " + a.text2bin_output);
    std::ofstream fcount(a.text2bin_count_output);
    uint64_t count = 0;
    std::string line;
    if (a.text2bin_mode == "variable") {          // src/text2bin.cc:19-47
      while (getline(fin, line)) {
        std::istringstream ss(line);
        uint64_t vid, role, init, type, card;
        if (!(ss >> vid >> role >> init >> type >> card)) throw std::runtime_error("bad variable line: " + line);
        put_be64(fout, vid); fout.put((char)(uint8_t)role); put_be64(fout, init);
        put_be16(fout, (uint16_t)type); put_be64(fout, card);
        ++count;
      }
    } else if (a.text2bin_mode == "weight") {     // :50-71
      while (getline(fin, line)) {
        std::istringstream ss(line);
        uint64_t wid, fixed; double val;
        if (!(ss >> wid >> fixed >> val)) throw std::runtime_error("bad weight line: " + line);
        put_be64(fout, wid); fout.put((char)(uint8_t)fixed); put_bef64(fout, val);
        ++count;
      }
    } else if (a.text2bin_mode == "factor") {     // :115-185; count = total edges
      const uint64_t ar = a.text2bin_factor_arity;
      const bool cat = a.text2bin_factor_func_id == 12;   // FUNC_AND_CATEGORICAL
      while (getline(fin, line)) {
        std::vector<std::string> f = split_tabs(line);
        const size_t need = ar + (cat ? ar : 0) + 2;
        if (f.size() < need) throw std::runtime_error("bad factor line: " + line);
        put_be16(fout, (uint16_t)a.text2bin_factor_func_id);
        put_be64(fout, ar);
        for (uint64_t i = 0; i < ar; ++i) {
          put_be64(fout, (uint64_t)atol(f[i].c_str()));
          uint64_t eq = cat ? (uint64_t)atol(f[ar + i].c_str())
                            : a.text2bin_factor_variables_should_equal_to.at(i);
          put_be64(fout, eq);
        }
        put_be64(fout, (uint64_t)atol(f[need - 2].c_str()));
        put_bef64(fout, atof(f[need - 1].c_str()));
        count += ar;
      }
    } else if (a.text2bin_mode == "domain") {     // :188-243
      while (getline(fin, line)) {
        std::istringstream ss(line);
        uint64_t vid, card; std::string dom, truthy;
        if (!(ss >> vid >> card >> dom >> truthy)) throw std::runtime_error("bad domain line: " + line);
        std::vector<std::string> dv = parse_pgarray(dom), tv = parse_pgarray(truthy);
        if (dv.size() != card || tv.size() != card) throw std::runtime_error("domain size != cardinality: " + line);
        put_be64(fout, vid); put_be64(fout, card);
        for (uint64_t i = 0; i < card; ++i) { put_be64(fout, (uint64_t)atol(dv[i].c_str())); put_bef64(fout, atof(tv[i].c_str())); }
        ++count;
      }
    } else {
      std::cerr << "Unsupported type" << std::endl;
      return 1;
    }
    fcount << count << std::endl;
    return 0;
  } catch (const std::exception &e) {
    std::cerr << "dw text2bin: " << e.what() << std::endl;
    return 1;
  }
}

int bin2text(const CmdLine &a) {
  try {
    LoadedGraph g;
    read_meta(a.fg_file, g);
    load_variables(a.variable_file, g);
    load_weights(a.weight_file, g);
    load_domains(a.domain_file, g);
    load_factors(a.factor_file, g);
    const std::string dir = a.output_folder;
    // per-variable domain (file order = dense order); implicit 0..card-1 without a block
    std::vector<int64_t> dom_of(g.n_variables, -1);
    for (size_t b = 0; b < g.dom_vid.size(); ++b) if (g.dom_vid[b] < g.n_variables) dom_of[g.dom_vid[b]] = (int64_t)b;
    auto sparse_value = [&](uint64_t vid, uint64_t dense) -> uint64_t {
      return dom_of[vid] >= 0 ? g.dom_value[g.dom_offset[dom_of[vid]] + dense] : dense;
    };
    auto dense_index = [&](uint64_t vid, uint64_t sparse) -> uint64_t {
      if (dom_of[vid] < 0) return sparse;
      for (uint64_t i = g.dom_offset[dom_of[vid]]; i < g.dom_offset[dom_of[vid] + 1]; ++i)
        if (g.dom_value[i] == sparse) return i - g.dom_offset[dom_of[vid]];
      throw std::runtime_error("value not in domain");
    };
    {   // dump_variables (src/bin2text.cc:23-50): prints assignment_dense
      std::ofstream f(dir + "/variables.tsv");
      for (uint64_t v = 0; v < g.n_variables; ++v) {
        const bool ev = g.var_role[v] >= 1;
        uint64_t val = ev ? g.var_init_value[v] : 0;
        if (val && dom_of[v] >= 0) val = dense_index(v, val);
        f << v << '\t' << (ev ? "1" : "0") << '\t' << val << '\t' << g.var_dtype[v] << '\t'
          << g.var_cardinality[v] << std::endl;
      }
    }
    {   // dump_domains (:52-69)
      std::ofstream f(dir + "/domains.tsv");
      for (uint64_t v = 0; v < g.n_variables; ++v) {
        if (g.var_dtype[v] == 0) continue;
        f << v << '\t' << g.var_cardinality[v] << '\t' << "{";
        for (uint64_t j = 0; j < g.var_cardinality[v]; ++j) f << (j ? "," : "") << sparse_value(v, j);
        f << "}" << std::endl;
      }
    }
    {   // dump_factors (:71-98)
      std::ofstream f(dir + "/factors.tsv");
      for (uint64_t i = 0; i < g.n_factors; ++i) {
        const uint64_t lo = g.fac_edge_offset[i], hi = g.fac_edge_offset[i + 1];
        for (uint64_t e = lo; e < hi; ++e) f << g.edge_vid[e] << '\t';
        if (g.fac_func[i] == 12) for (uint64_t e = lo; e < hi; ++e) f << g.edge_equal_to[e] << '\t';
        f << g.fac_weight_id[i] << '\t' << g.fac_feature_value[i] << std::endl;
      }
    }
    {   // dump_weights (:100-112)
      std::ofstream f(dir + "/weights.tsv");
      for (uint64_t w = 0; w < g.n_weights; ++w)
        f << w << '\t' << (int)g.w_is_fixed[w] << '\t' << g.w_initial_value[w] << std::endl;
    }
    {   // dump_meta (:114-129)
      std::ofstream f(dir + "/graph.meta");
      f << g.n_weights << "," << g.n_variables << "," << g.n_factors << "," << g.n_edges
        << ",graph.weights,graph.variables,graph.factors" << std::endl;
    }
    return 0;
  } catch (const std::exception &e) {
    std::cerr << "dw bin2text: " << e.what() << std::endl;
    return 1;
  }
}

int dw_main(int argc, const char *const argv[]) {
  CmdLine a = parse_cmdline(argc, argv);
  if (a.num_errors > 0) {
    std::cerr << a.error_text;
    return a.num_errors;
  }
  if (a.app_name == "text2bin") return text2bin(a);
  if (a.app_name == "bin2text") return bin2text(a);
  if (a.gpus >= 1 || a.n_datacopy > 1) return gibbs_multi(a);
  return gibbs(a);
}

}  // namespace dw
