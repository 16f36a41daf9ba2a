// hdf5_io.h -- the three datasets of the reference's Criteo file (X_int, X_cat, y) read through libhdf5,
// which is dlopen'ed at run time: the build has no link-time HDF5 dependency and a machine without the
// library can still run the synthetic path.  [ref: examples/cpp/DLRM/dlrm.cc:279-326,421-479 (the H5* calls this
// replaces); file layout: examples/cpp/DLRM/preprocess_hdf.py:14-24]
#pragma once
#include <cstdint>
#include <string>
#include <vector>

struct Hdf5Dataset {
  std::vector<uint64_t> dims;   // extent
  int type_class;               // 0 = H5T_INTEGER, 1 = H5T_FLOAT
};

class Hdf5File {
 public:
  // opens read-only; aborts with a message when libhdf5 or the file cannot be opened
  explicit Hdf5File(const std::string& path);
  ~Hdf5File();
  Hdf5Dataset describe(const char* name);
  // rows [row0, row0 + nrows) of a 1-D or 2-D dataset, converted by the library to float / int64
  void read_rows_f32(const char* name, uint64_t row0, uint64_t nrows, float* out);
  void read_rows_i64(const char* name, uint64_t row0, uint64_t nrows, int64_t* out);
  static std::string library_path();   // the libhdf5 in use ("" before the first open)

 private:
  void read_rows(const char* name, uint64_t row0, uint64_t nrows, void* out, bool as_float);
  int64_t file_id;
  std::string path;
};
