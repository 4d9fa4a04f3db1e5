// The netCDF classic format (CDF-2, "64-bit offset": what fms_io and diag_manager write by default), written without a netCDF library: dimensions,
// text attributes, double variables, one record dimension.  Shared by the restart files (restart_nc.cpp) and the history files (history_nc.cpp).
// Two ways to write: write(path, numrecs) -- everything at once, the records fetched one by one through each variable's fill(record, out) --, or
// begin(path) / append() / finish(): the header and the fixed variables first, a record at a time as the run produces them, the record count patched
// into the header at the end (a history file grows with the run).
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <stdexcept>
#include <string>
#include <vector>

namespace isca_nc3 {

[[noreturn]] inline void nc3_fail(const std::string &m) { throw std::runtime_error(m); }

enum { NC_BYTE = 1, NC_CHAR = 2, NC_SHORT = 3, NC_INT = 4, NC_FLOAT = 5, NC_DOUBLE = 6, NC_DIMENSION = 10, NC_VARIABLE = 11, NC_ATTRIBUTE = 12 };

inline uint64_t bswap64(uint64_t x) { return __builtin_bswap64(x); }
inline uint32_t bswap32(uint32_t x) { return __builtin_bswap32(x); }
inline size_t pad4(size_t n) { return (n + 3) & ~(size_t)3; }

// ------------------------------------------------------------------------------------------------ writer
struct WVar {
  std::string name;
  std::vector<int> dims;                                       // dimension ids, slowest first (the record dimension, if any, first)
  std::vector<std::pair<std::string, std::string>> atts;      // text attributes
  bool rec = false;
  size_t count = 0;                                            // doubles per record (record variable) or in total
  std::function<void(int, double *)> fill;                     // (record or -1, out[count])
  uint64_t begin = 0;
};
class Nc3Writer {
 public:
  int dim(const std::string &name, size_t len) {               // len 0: the record dimension
    for (size_t i = 0; i < dims_.size(); ++i) if (dims_[i].first == name) return (int)i;
    dims_.push_back({name, len});
    return (int)dims_.size() - 1;
  }
  size_t dim_len(int id) const { return dims_[id].second; }
  void var(const std::string &name, const std::vector<int> &dims, std::vector<std::pair<std::string, std::string>> atts,
           std::function<void(int, double *)> fill) {
    WVar v; v.name = name; v.dims = dims; v.atts = std::move(atts); v.fill = std::move(fill);
    v.rec = !dims.empty() && dims_[dims[0]].second == 0;
    v.count = 1;
    for (size_t i = v.rec ? 1 : 0; i < dims.size(); ++i) v.count *= dims_[dims[i]].second;
    vars_.push_back(std::move(v));
  }
  void write(const std::string &path, int numrecs) {
    // header size with 64-bit begins, then the offsets: fixed variables in definition order, then the records (every record variable's
    // slab, in definition order, per record)
    std::vector<unsigned char> hd;
    uint64_t off = header(hd, numrecs, false);
    for (auto &v : vars_) if (!v.rec) { v.begin = off; off += v.count * 8; }
    uint64_t recsize = 0;
    for (auto &v : vars_) if (v.rec) { v.begin = off + recsize; recsize += v.count * 8; }
    hd.clear();
    header(hd, numrecs, true);
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) nc3_fail("write_data: cannot open " + path);
    bool ok = fwrite(hd.data(), 1, hd.size(), f) == hd.size();
    std::vector<double> buf;
    std::vector<uint64_t> be;
    auto put = [&](WVar &v, int rec) {
      buf.resize(v.count); be.resize(v.count);
      v.fill(rec, buf.data());
      for (size_t i = 0; i < v.count; ++i) { uint64_t u; memcpy(&u, &buf[i], 8); be[i] = bswap64(u); }
      ok = ok && fwrite(be.data(), 8, v.count, f) == v.count;
    };
    for (auto &v : vars_) if (!v.rec) put(v, -1);
    for (int r = 0; r < numrecs; ++r)
      for (auto &v : vars_) if (v.rec) put(v, r);
    ok = (fclose(f) == 0) && ok;
    if (!ok) nc3_fail("write_data: error writing " + path);
  }
  // ---- a record at a time
  void begin(const std::string &path) {                         // header with a record count of 0 + the fixed variables
    layout(0);
    path_ = path;
    f_ = fopen(path.c_str(), "wb");
    if (!f_) nc3_fail("write_data: cannot open " + path);
    std::vector<unsigned char> hd;
    header(hd, 0, true);
    bool ok = fwrite(hd.data(), 1, hd.size(), f_) == hd.size();
    for (auto &v : vars_) if (!v.rec) ok = put(v, -1) && ok;
    if (!ok || fflush(f_) != 0) nc3_fail("write_data: error writing " + path_);
    nrec_ = 0;
  }
  void append() {                                              // record nrec_: every record variable's slab, in definition order
    if (!f_) nc3_fail("write_data: append() on a file that is not open");
    bool ok = true;
    for (auto &v : vars_) if (v.rec) ok = put(v, nrec_) && ok;
    ++nrec_;
    unsigned char n4[4] = {(unsigned char)(nrec_ >> 24), (unsigned char)(nrec_ >> 16), (unsigned char)(nrec_ >> 8), (unsigned char)nrec_};
    ok = ok && fseeko(f_, 4, SEEK_SET) == 0 && fwrite(n4, 1, 4, f_) == 4 && fseeko(f_, 0, SEEK_END) == 0 && fflush(f_) == 0;   // the count in the header: the file is complete after every record
    if (!ok) nc3_fail("write_data: error writing " + path_);
  }
  int records() const { return nrec_; }
  bool is_open() const { return f_ != nullptr; }
  void finish() {
    if (f_ && fclose(f_) != 0) { f_ = nullptr; nc3_fail("write_data: error closing " + path_); }
    f_ = nullptr;
  }
  ~Nc3Writer() { if (f_) fclose(f_); }
  Nc3Writer() = default;
  Nc3Writer(const Nc3Writer &) = delete;
  Nc3Writer &operator=(const Nc3Writer &) = delete;

 private:
  void layout(int numrecs) {
    std::vector<unsigned char> hd;
    uint64_t off = header(hd, numrecs, false);
    for (auto &v : vars_) if (!v.rec) { v.begin = off; off += v.count * 8; }
    uint64_t recsize = 0;
    for (auto &v : vars_) if (v.rec) { v.begin = off + recsize; recsize += v.count * 8; }
  }
  bool put(WVar &v, int rec) {
    buf_.resize(v.count); be_.resize(v.count);
    v.fill(rec, buf_.data());
    for (size_t i = 0; i < v.count; ++i) { uint64_t u; memcpy(&u, &buf_[i], 8); be_[i] = bswap64(u); }
    return fwrite(be_.data(), 8, v.count, f_) == v.count;
  }
  FILE *f_ = nullptr;
  std::string path_;
  int nrec_ = 0;
  std::vector<double> buf_;
  std::vector<uint64_t> be_;
  static void put32(std::vector<unsigned char> &b, uint32_t x) { for (int s = 24; s >= 0; s -= 8) b.push_back((unsigned char)(x >> s)); }
  static void put64(std::vector<unsigned char> &b, uint64_t x) { for (int s = 56; s >= 0; s -= 8) b.push_back((unsigned char)(x >> s)); }
  static void putname(std::vector<unsigned char> &b, const std::string &s) {
    put32(b, (uint32_t)s.size());
    b.insert(b.end(), s.begin(), s.end());
    while (b.size() & 3) b.push_back(0);
  }
  uint64_t header(std::vector<unsigned char> &b, int numrecs, bool with_begins) {
    b.push_back('C'); b.push_back('D'); b.push_back('F'); b.push_back(2);
    put32(b, (uint32_t)numrecs);
    put32(b, NC_DIMENSION); put32(b, (uint32_t)dims_.size());
    for (auto &d : dims_) { putname(b, d.first); put32(b, (uint32_t)d.second); }
    put32(b, 0); put32(b, 0);                                   // no global attributes
    put32(b, NC_VARIABLE); put32(b, (uint32_t)vars_.size());
    for (auto &v : vars_) {
      putname(b, v.name);
      put32(b, (uint32_t)v.dims.size());
      for (int d : v.dims) put32(b, (uint32_t)d);
      if (v.atts.empty()) { put32(b, 0); put32(b, 0); }
      else {
        put32(b, NC_ATTRIBUTE); put32(b, (uint32_t)v.atts.size());
        for (auto &a : v.atts) { putname(b, a.first); put32(b, NC_CHAR); putname(b, a.second); }
      }
      put32(b, NC_DOUBLE);
      const uint64_t vsize = v.count * 8;
      put32(b, vsize > 0xfffffffcULL ? 0xffffffffu : (uint32_t)vsize);
      put64(b, with_begins ? v.begin : 0);
    }
    return b.size();
  }
  std::vector<std::pair<std::string, size_t>> dims_;
  std::vector<WVar> vars_;
};


}  // namespace isca_nc3
