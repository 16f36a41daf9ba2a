// strategy.cc -- ParallelConfig and the strategy text files of the reference
// [ref: src/runtime/strategy.cc:22-189, include/config.h:47-73].  A file written by the reference's
// save_strategies_to_file (or by its search, --export) loads here unchanged, and the file this build
// writes loads in the reference.  What a config may ask for is checked where it is applied
// (FFModel::apply_strategies, model.cc).
#include "ffmodel.h"

#include <cstdio>
#include <fstream>
#include <functional>

MappingTagID FFConfig::get_hash_id(const std::string& pcname) { return std::hash<std::string>{}(pcname); }

// The reference falls back to the DataParallelism_GPU_<n>D default when the name is absent
// [ref: strategy.cc:32-84]; here "absent" is reported and the caller keeps its own default placement.
bool FFConfig::find_parallel_config(int ndims, const std::string& pcname, ParallelConfig& config) const {
  auto it = strategies.find(get_hash_id(pcname));
  if (it == strategies.end()) return false;
  config = it->second;
  if (config.nDims != ndims) {
    fprintf(stderr, "FATAL: strategy for %s has %d dims, the op's output has %d\n", pcname.c_str(), config.nDims, ndims);
    abort();
  }
  return true;
}

bool ParallelConfig::is_data_parallel() const {
  for (int i = 0; i + 1 < nDims; i++)
    if (dim[i] != 1) return false;
  return true;
}

bool ParallelConfig::operator==(const ParallelConfig& rhs) const {
  if (nDims != rhs.nDims || device_type != rhs.device_type) return false;
  for (int i = 0; i < nDims; i++)
    if (dim[i] != rhs.dim[i]) return false;
  return device_ids == rhs.device_ids;
}

bool load_strategies_from_file(const std::string& filename, std::map<MappingTagID, ParallelConfig>& strategies) {
  std::fstream input(filename, std::ios::in);
  if (!input) {
    fprintf(stderr, "Failed to open strategy file for reading\n");
    return false;
  }
  auto bad = [&](const char* what, const std::string& op) {
    fprintf(stderr, "FATAL: strategy file %s: %s (op '%s')\n", filename.c_str(), what, op.c_str());
    abort();
  };
  int ops_size = 0;
  if (!(input >> ops_size) || ops_size < 0) bad("missing op count", "");
  for (int i = 0; i < ops_size; i++) {
    ParallelConfig config;
    std::string op_name;
    int device_type = -1;
    if (!(input >> op_name >> device_type)) bad("truncated entry", op_name);
    if (op_name.size() >= MAX_OPNAME) bad("op name too long", op_name);
    if (device_type != ParallelConfig::GPU && device_type != ParallelConfig::CPU) bad("Unsupported Device Type", op_name);
    config.device_type = (ParallelConfig::DeviceType)device_type;
    if (!(input >> config.nDims) || config.nDims < 1 || config.nDims > MAX_TENSOR_DIM) bad("nDims out of range", op_name);
    int n = 1;
    for (int j = 0; j < config.nDims; j++) {
      if (!(input >> config.dim[j]) || config.dim[j] < 1) bad("bad dim", op_name);
      n *= config.dim[j];
    }
    int ids = 0;
    if (!(input >> ids)) bad("missing device id count", op_name);
    if (ids != n && ids != 0) bad("device id count does not match the product of the dims", op_name);   // [ref: strategy.cc:135]
    config.device_ids.resize(ids);
    for (int j = 0; j < ids; j++)
      if (!(input >> config.device_ids[j])) bad("truncated device ids", op_name);
    const MappingTagID hash = FFConfig::get_hash_id(op_name);
    if (strategies.find(hash) != strategies.end()) bad("duplicate op", op_name);                           // [ref: strategy.cc:142]
    strategies[hash] = config;
  }
  printf("strategies.size() = %zu\n", strategies.size());
  return true;
}

bool save_strategies_to_file(const std::string& filename, const std::map<std::string, ParallelConfig>& strategies) {
  std::fstream output(filename, std::ios::out | std::ios::trunc);
  if (!output) {
    fprintf(stderr, "Failed to open strategy file for writing!\n");
    return false;
  }
  output << strategies.size() << std::endl;
  for (const auto& it : strategies) {
    const ParallelConfig& config = it.second;
    output << it.first << std::endl;
    output << (int)config.device_type << std::endl;
    output << config.nDims << std::endl;
    for (int j = 0; j < config.nDims; j++) output << config.dim[j] << '\t';
    output << std::endl;
    const int n = config.num_parts();
    output << n << std::endl;
    for (int j = 0; j < n; j++) output << (j < (int)config.device_ids.size() ? config.device_ids[j] : j) << '\t';
    output << std::endl;
  }
  return true;
}
