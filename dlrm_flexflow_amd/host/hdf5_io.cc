// hdf5_io.cc -- see hdf5_io.h.  Only the C API of HDF5 >= 1.10 (64-bit hid_t) is used.
#include "hdf5_io.h"

#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>

namespace {

typedef int64_t hid_t;
typedef int herr_t;
typedef unsigned long long hsize_t;

struct H5 {
  void* lib = nullptr;
  std::string path;
  herr_t (*open)(void);
  herr_t (*get_libversion)(unsigned*, unsigned*, unsigned*);
  hid_t (*Fopen)(const char*, unsigned, hid_t);
  herr_t (*Fclose)(hid_t);
  hid_t (*Dopen2)(hid_t, const char*, hid_t);
  herr_t (*Dclose)(hid_t);
  hid_t (*Dget_space)(hid_t);
  hid_t (*Dget_type)(hid_t);
  herr_t (*Dread)(hid_t, hid_t, hid_t, hid_t, hid_t, void*);
  int (*Sget_simple_extent_ndims)(hid_t);
  int (*Sget_simple_extent_dims)(hid_t, hsize_t*, hsize_t*);
  hid_t (*Screate_simple)(int, const hsize_t*, const hsize_t*);
  herr_t (*Sselect_hyperslab)(hid_t, int, const hsize_t*, const hsize_t*, const hsize_t*, const hsize_t*);
  herr_t (*Sclose)(hid_t);
  int (*Tget_class)(hid_t);
  herr_t (*Tclose)(hid_t);
  herr_t (*Eset_auto2)(hid_t, void*, void*);
  hid_t native_float, native_llong;
};

[[noreturn]] void die(const char* fmt, const std::string& a, const std::string& b = "") {
  fprintf(stderr, "FATAL: ");
  fprintf(stderr, fmt, a.c_str(), b.c_str());
  fprintf(stderr, "\n");
  abort();
}

H5& h5() {
  static H5 h;
  if (h.lib) return h;
  std::vector<std::string> names;
  if (const char* e = getenv("FFH_HDF5_LIB")) names.push_back(e);
  else
    for (const char* n : {"libhdf5.so", "libhdf5_serial.so", "libhdf5.so.103", "libhdf5_serial.so.103", "libhdf5.so.200", "libhdf5.so.310",
                          "/opt/conda/lib/libhdf5.so"})
      names.push_back(n);
  std::string tried;
  for (const std::string& n : names) {
    h.lib = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (h.lib) { h.path = n; break; }
    tried += " " + n;
  }
  if (!h.lib) die("--dataset needs libhdf5 (>= 1.10) at run time; tried:%s (set FFH_HDF5_LIB to its path)", tried);
#define SYM(field, name)                                                         \
  *(void**)(&h.field) = dlsym(h.lib, name);                                      \
  if (!h.field) die("%s lacks the symbol %s", h.path, name);
  SYM(open, "H5open") SYM(get_libversion, "H5get_libversion") SYM(Fopen, "H5Fopen") SYM(Fclose, "H5Fclose")
  SYM(Dopen2, "H5Dopen2") SYM(Dclose, "H5Dclose") SYM(Dget_space, "H5Dget_space") SYM(Dget_type, "H5Dget_type") SYM(Dread, "H5Dread")
  SYM(Sget_simple_extent_ndims, "H5Sget_simple_extent_ndims") SYM(Sget_simple_extent_dims, "H5Sget_simple_extent_dims")
  SYM(Screate_simple, "H5Screate_simple") SYM(Sselect_hyperslab, "H5Sselect_hyperslab") SYM(Sclose, "H5Sclose")
  SYM(Tget_class, "H5Tget_class") SYM(Tclose, "H5Tclose") SYM(Eset_auto2, "H5Eset_auto2")
#undef SYM
  unsigned maj = 0, min = 0, rel = 0;
  h.get_libversion(&maj, &min, &rel);
  if (maj < 1 || (maj == 1 && min < 10)) die("%s is older than HDF5 1.10 (32-bit handles)%s", h.path);
  if (h.open() < 0) die("H5open failed in %s%s", h.path);
  hid_t* f = (hid_t*)dlsym(h.lib, "H5T_NATIVE_FLOAT_g");
  hid_t* l = (hid_t*)dlsym(h.lib, "H5T_NATIVE_LLONG_g");
  if (!f || !l) die("%s lacks the native type ids%s", h.path);
  h.native_float = *f;
  h.native_llong = *l;
  h.Eset_auto2(0 /*H5E_DEFAULT*/, nullptr, nullptr);   // errors are reported here, with the file and dataset name
  return h;
}

}  // namespace

std::string Hdf5File::library_path() { return h5().path; }

Hdf5File::Hdf5File(const std::string& p) : file_id(-1), path(p) {
  H5& h = h5();
  file_id = h.Fopen(p.c_str(), 0u /*H5F_ACC_RDONLY*/, 0 /*H5P_DEFAULT*/);
  if (file_id < 0) die("cannot open %s as an HDF5 file%s", p);
}

Hdf5File::~Hdf5File() {
  if (file_id >= 0) h5().Fclose(file_id);
}

Hdf5Dataset Hdf5File::describe(const char* name) {
  H5& h = h5();
  const hid_t d = h.Dopen2(file_id, name, 0);
  if (d < 0) die("%s has no dataset '%s'", path, name);
  const hid_t sp = h.Dget_space(d), ty = h.Dget_type(d);
  Hdf5Dataset out;
  const int nd = h.Sget_simple_extent_ndims(sp);
  if (nd < 1 || nd > 2) die("%s: dataset '%s' must have 1 or 2 dimensions", path, name);
  hsize_t dims[2] = {0, 0}, maxdims[2];
  h.Sget_simple_extent_dims(sp, dims, maxdims);
  for (int i = 0; i < nd; i++) out.dims.push_back(dims[i]);
  out.type_class = h.Tget_class(ty);
  h.Tclose(ty); h.Sclose(sp); h.Dclose(d);
  return out;
}

void Hdf5File::read_rows(const char* name, uint64_t row0, uint64_t nrows, void* out, bool as_float) {
  if (nrows == 0) return;
  H5& h = h5();
  const hid_t d = h.Dopen2(file_id, name, 0);
  if (d < 0) die("%s has no dataset '%s'", path, name);
  const hid_t sp = h.Dget_space(d);
  const int nd = h.Sget_simple_extent_ndims(sp);
  hsize_t dims[2] = {0, 1}, maxdims[2];
  h.Sget_simple_extent_dims(sp, dims, maxdims);
  if (row0 + nrows > dims[0]) die("%s: read past the end of '%s'", path, name);
  const hsize_t start[2] = {row0, 0}, count[2] = {nrows, nd == 2 ? dims[1] : 1};
  if (h.Sselect_hyperslab(sp, 0 /*H5S_SELECT_SET*/, start, nullptr, count, nullptr) < 0) die("%s: hyperslab selection on '%s' failed", path, name);
  const hid_t mem = h.Screate_simple(nd, count, nullptr);
  const herr_t rc = h.Dread(d, as_float ? h.native_float : h.native_llong, mem, sp, 0, out);
  h.Sclose(mem); h.Sclose(sp); h.Dclose(d);
  if (rc < 0) die("%s: H5Dread of '%s' failed", path, name);
}

void Hdf5File::read_rows_f32(const char* name, uint64_t row0, uint64_t nrows, float* out) { read_rows(name, row0, nrows, out, true); }
void Hdf5File::read_rows_i64(const char* name, uint64_t row0, uint64_t nrows, int64_t* out) { read_rows(name, row0, nrows, out, false); }
