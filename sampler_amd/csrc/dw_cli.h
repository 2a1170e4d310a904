// dw_cli.h -- host side of the `dw` drop-in binary: command line, big-endian loader,
// epoch driver and result dumps, all in C++ above the C ABI (include/dwx.h), mirroring
// the reference's L4/L5/L0 layers:
//   CmdParser              /root/reference/src/cmd_parser.cc:44-259
//   read_meta / load_*     /root/reference/src/binary_format.cc:23-226
//   gibbs(), DimmWitted    /root/reference/src/dimmwitted.cc:37-282
//   dump_*_in_text         /root/reference/src/inference_result.cc:101-105,211-243
#ifndef DWX_DW_CLI_H_
#define DWX_DW_CLI_H_

#include <stdint.h>

#include <iosfwd>
#include <string>
#include <vector>

#include "../../include/dwx.h"
#include "host_parallel.h"

namespace dw {

// ---- command line (same flags, defaults and validation messages as the reference)
struct CmdLine {
  std::string app_name;
  std::string fg_file;
  std::vector<std::string> variable_file, domain_file, factor_file, weight_file;
  std::string output_folder;
  uint64_t n_learning_epoch = 0, n_inference_epoch = 0, n_datacopy = 0, n_threads = 0, burn_in = 0;
  double stepsize = 0.01, stepsize2 = 0.01, decay = 0.95, reg_param = 0.01;
  bool regularization_l1 = false;
  bool should_be_quiet = false, should_sample_evidence = false, should_learn_non_evidence = false,
       is_noise_aware = false;
  // text2bin (src/cmd_parser.cc:168-212)
  std::string text2bin_mode, text2bin_input, text2bin_output, text2bin_count_output;
  int text2bin_factor_func_id = 0;
  uint64_t text2bin_factor_arity = 1;
  std::vector<uint64_t> text2bin_factor_variables_should_equal_to;
  // additions of this build
  int device = 0;
  uint64_t seed = 0x5eed5eedULL;
  double step_cap = 1.5;
  int plan_layouts = 0;        // dwx_options.plan_layouts (0: after 2048 sweeps, 1: at once, 2: never)
  int gpus = 0;                 // --gpus N: variable-block shards over N GPUs (dw_multi.h)
  std::vector<int> devices;     // --devices a,b,...: the ranks' HIP devices (default 0..N-1)
  std::string comm = "rccl";    // --comm rccl | host (host-staged sums: a test stand-in)
  int num_errors = 0;
  std::string error_text;
};

// Parses argv (argv[0] = program, argv[1] = mode).  Errors are counted and described
// in error_text, like CmdParser::check (src/cmd_parser.cc:31-38).
CmdLine parse_cmdline(int argc, const char *const argv[]);
std::ostream &operator<<(std::ostream &o, const CmdLine &a);  // src/cmd_parser.cc:266-291

// ---- loader: the reference's binary files -> columnar arrays (dwx_graph_desc)
struct LoadedGraph {
  uint64_t n_weights = 0, n_variables = 0, n_factors = 0, n_edges = 0;
  uint64_t n_evidence = 0, n_query = 0;
  std::vector<uint8_t> var_role;
  std::vector<uint64_t> var_init_value, var_cardinality;
  std::vector<uint16_t> var_dtype;
  std::vector<uint64_t> dom_vid, dom_offset, dom_value;
  std::vector<double> dom_truthiness;
  // the big columns are filled by parallel decoders: no serial zero-fill
  dwx::RawArray<uint16_t> fac_func;
  dwx::RawArray<uint64_t> fac_edge_offset, fac_weight_id;
  dwx::RawArray<double> fac_feature_value;
  dwx::RawArray<uint64_t> edge_vid, edge_equal_to;
  std::vector<double> w_initial_value;
  std::vector<uint8_t> w_is_fixed;
  dwx_graph_desc desc() const;
};

// Throws std::runtime_error on malformed files (the reference assert()s).
void read_meta(const std::string &path, LoadedGraph &g);
void load_variables(const std::vector<std::string> &files, LoadedGraph &g);
void load_weights(const std::vector<std::string> &files, LoadedGraph &g);
void load_domains(const std::vector<std::string> &files, LoadedGraph &g);
void load_factors(const std::vector<std::string> &files, LoadedGraph &g);

// ---- result dumps
void dump_weights_in_text(std::ostream &o, const std::vector<double> &w);
void dump_marginals_in_text(std::ostream &o, const LoadedGraph &g, bool sample_evidence,
                            const uint64_t *var_val_base, const uint64_t *value_sparse,
                            const uint64_t *tallies, const uint64_t *nsamples,
                            uint64_t id_offset = 0, uint64_t n_vars = ~0ull);   // (a shard: local ids + offset, owned only)

// the same two dumps straight into a file (the single-GPU `dw gibbs`)
void dump_weights_to_file(const std::string &path, const std::vector<double> &w);
void dump_marginals_to_file(const std::string &path, const LoadedGraph &g, bool sample_evidence,
                            const uint64_t *var_val_base, const uint64_t *value_sparse,
                            const uint64_t *tallies, const uint64_t *nsamples);

// graph-compile options of a run (dw_cli.cc: the weight order of the variables only for long runs)
dwx_compile_opts compile_opts_for(const CmdLine &args);
// flush and leave the process with exit_code unless DWX_FULL_TEARDOWN is set (dw_cli.cc)
void quick_exit_if_done(int exit_code);
// the `dw gibbs` mode (src/dimmwitted.cc:37-95); returns the process exit code
int gibbs(const CmdLine &args);
// the same over several GPUs (dw_multi.cc): --gpus N shards, -c N replicas
int gibbs_multi(const CmdLine &args);
// `dw text2bin` (src/text2bin.cc:19-260) and `dw bin2text` (src/bin2text.cc:23-151)
int text2bin(const CmdLine &args);
int bin2text(const CmdLine &args);
// entry point (src/dimmwitted.cc:20-35)
int dw_main(int argc, const char *const argv[]);

}  // namespace dw
#endif
