// Restart files from the library itself: spectral_dynamics.res.nc, atmosphere.res.nc and (moist package) mixed_layer.res.nc with the
// reference's variable set, written and read in the netCDF classic format (CDF-1 / CDF-2 "64-bit offset", the format fms_io writes by
// default) by the few hundred lines below -- no netCDF library.  Replaces, for the Fortran host behind bindings/fortran/dropin, what
// fms_io's write_data / read_data / field_size do for
//   spectral_dynamics_end          (src/atmos_spectral/model/spectral_dynamics.F90:1502-1531)
//   read_restart_or_do_coldstart   (spectral_dynamics.F90:509-575)
//   atmosphere_end / atmosphere_init (src/atmos_spectral/driver/solo/atmosphere.F90:362-375, 197-223)
//   mixed_layer_end / _init        (src/atmos_spectral/driver/solo/mixed_layer.F90:813, 324-327)
// The Python host mirror (isca_amd/restart.py, scipy.io.netcdf_file) writes and reads the same files: each side reads what the other
// wrote (tests/test_gpu_parity.py::test_native_restart_files), and a run continued from them equals the uninterrupted one bit for bit.
// Layout of a file as fms_io leaves it: every field is (Time, zaxis_k, yaxis_j, xaxis_i) with coordinate variables xaxis_N / yaxis_N /
// zaxis_N = 1..n (attribute cartesian_axis) and the record dimension Time; the reader goes by variable NAME and takes whatever
// dimensions it finds (read_data / field_size do the same).
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <algorithm>
#include <stdexcept>
#include <string>
#include <vector>
#include <sys/stat.h>
#include "core.h"
#include "nc3.h"

void isca_internal_set_error(const std::string &m);       // api.hip: the thread's isca_last_error() text

namespace {

[[noreturn]] void fail(const std::string &m) { throw std::runtime_error(m); }

using namespace isca_nc3;

// One fms_io-style restart file: every variable (Time, zaxis, yaxis, xaxis), axes created on first use (isca_amd/restart.py: _Writer)
class RestartFile {
 public:
  RestartFile() { time_ = w_.dim("Time", 0); }
  // nrec records of shape (z, y, x); fill(record, out)
  void put(const std::string &name, int nrec, size_t z, size_t y, size_t x, std::function<void(int, double *)> fill) {
    const std::vector<int> dims = {time_, axis('z', z), axis('y', y), axis('x', x)};
    // a variable with fewer records than the file repeats its last one (the classic format has one record count per file)
    w_.var(name, dims, {}, [fill, nrec](int r, double *out) { fill(r < nrec ? r : nrec - 1, out); });
    nrec_ = std::max(nrec_, nrec);
  }
  void close(const std::string &path) {
    w_.var("Time", {time_}, {{"cartesian_axis", "T"}}, [](int r, double *out) { out[0] = r + 1.0; });
    w_.write(path, nrec_);
  }

 private:
  int axis(char kind, size_t n) {
    auto &sizes = axes_[kind];
    size_t k = 0;
    while (k < sizes.size() && sizes[k] != n) ++k;
    const std::string name = std::string(1, kind) + "axis_" + std::to_string(k + 1);
    if (k == sizes.size()) {
      sizes.push_back(n);
      const int id = w_.dim(name, n);
      w_.var(name, {id}, {{"cartesian_axis", std::string(1, (char)(kind - 32))}}, [n](int, double *out) { for (size_t i = 0; i < n; ++i) out[i] = i + 1.0; });
    }
    return w_.dim(name, n);
  }
  Nc3Writer w_;
  int time_, nrec_ = 0;
  std::map<char, std::vector<size_t>> axes_;
};

// ------------------------------------------------------------------------------------------------ reader
struct RVar {
  std::vector<size_t> shape;     // dimension lengths, slowest first; a record variable's first entry is the file's record count
  bool rec = false;
  int type = NC_DOUBLE;
  uint64_t begin = 0;
  size_t count = 0;              // values per record (record variable) or in total
};
class Nc3File {
 public:
  explicit Nc3File(const std::string &path) : path_(path) {
    f_ = fopen(path.c_str(), "rb");
    if (!f_) fail("read_data: cannot open " + path);
    try { parse_header(); }
    catch (...) { fclose(f_); f_ = nullptr; throw; }          // (a constructor that throws runs no destructor: a malformed file must not leak its handle)
  }
  void parse_header() {
    unsigned char magic[4];
    need(fread(magic, 1, 4, f_) == 4 && magic[0] == 'C' && magic[1] == 'D' && magic[2] == 'F' && (magic[3] == 1 || magic[3] == 2),
         "is not a netCDF classic / 64-bit-offset file (a netCDF-4 file must be converted: nccopy -k 64-bit-offset)");
    const bool big = magic[3] == 2;
    numrecs_ = get32();
    std::vector<size_t> dimlen;
    uint32_t tag = get32(), n = get32();
    need(tag == NC_DIMENSION || (tag == 0 && n == 0), "has a damaged dimension list");
    for (uint32_t i = 0; i < n; ++i) { getname(); dimlen.push_back(get32()); }
    skip_atts();
    tag = get32(); n = get32();
    need(tag == NC_VARIABLE || (tag == 0 && n == 0), "has a damaged variable list");
    uint64_t recsize = 0;
    for (uint32_t i = 0; i < n; ++i) {
      const std::string name = getname();
      RVar v;
      const uint32_t nd = get32();
      for (uint32_t d = 0; d < nd; ++d) {
        const uint32_t id = get32();
        need(id < dimlen.size(), "has a variable with an unknown dimension");
        if (d == 0 && dimlen[id] == 0) { v.rec = true; v.shape.push_back(numrecs_); }
        else v.shape.push_back(dimlen[id]);
      }
      skip_atts();
      v.type = (int)get32();
      get32();                                                  // vsize (recomputed)
      v.begin = big ? get64() : get32();
      v.count = 1;
      for (size_t d = v.rec ? 1 : 0; d < v.shape.size(); ++d) v.count *= v.shape[d];
      if (v.rec) { recsize += pad4(v.count * tsize(v.type)); ++nrecvars_; last_rec_bytes_ = v.count * tsize(v.type); }
      vars_[name] = v;
    }
    recsize_ = (nrecvars_ == 1) ? last_rec_bytes_ : recsize;   // a single record variable is not padded (the format's special case)
  }
  ~Nc3File() { if (f_) fclose(f_); }
  bool has(const std::string &name) const { return vars_.count(name) != 0; }
  const RVar &var(const std::string &name) const {
    auto it = vars_.find(name);
    if (it == vars_.end()) fail("read_data: " + path_ + " has no variable " + name);
    return it->second;
  }
  // the values of record `rec` (record variable: 0-based, clamped to the last one present) or the whole (fixed) variable
  std::vector<double> read(const std::string &name, int rec) {
    const RVar &v = var(name);
    uint64_t off = v.begin;
    if (v.rec) {
      if (numrecs_ == 0) fail("read_data: " + path_ + ": " + name + " has no records");
      off += (uint64_t)std::max<long>(0, std::min<long>(rec, (long)numrecs_ - 1)) * recsize_;      // (a negative record: the first)
    }
    const size_t ts = tsize(v.type);
    std::vector<unsigned char> raw(v.count * ts);
    need(fseeko(f_, (off_t)off, SEEK_SET) == 0 && fread(raw.data(), 1, raw.size(), f_) == raw.size(), "ends inside " + name);
    std::vector<double> out(v.count);
    for (size_t i = 0; i < v.count; ++i) {
      const unsigned char *p = raw.data() + i * ts;
      switch (v.type) {
        case NC_DOUBLE: { uint64_t u; memcpy(&u, p, 8); u = bswap64(u); double x; memcpy(&x, &u, 8); out[i] = x; break; }
        case NC_FLOAT: { uint32_t u; memcpy(&u, p, 4); u = bswap32(u); float x; memcpy(&x, &u, 4); out[i] = x; break; }
        case NC_INT: { uint32_t u; memcpy(&u, p, 4); out[i] = (double)(int32_t)bswap32(u); break; }
        case NC_SHORT: out[i] = (double)(int16_t)((p[0] << 8) | p[1]); break;
        default: out[i] = (double)(signed char)p[0];
      }
    }
    return out;
  }

 private:
  static size_t tsize(int t) { return t == NC_DOUBLE ? 8 : (t == NC_FLOAT || t == NC_INT) ? 4 : t == NC_SHORT ? 2 : 1; }
  void need(bool ok, const std::string &what) const { if (!ok) fail("read_data: " + path_ + " " + what); }
  uint32_t get32() { unsigned char b[4]; need(fread(b, 1, 4, f_) == 4, "ends inside its header"); return ((uint32_t)b[0] << 24) | (b[1] << 16) | (b[2] << 8) | b[3]; }
  uint64_t get64() { const uint64_t hi = get32(); return (hi << 32) | get32(); }
  std::string getname() {
    const uint32_t n = get32();
    need(n < 4096, "has a damaged name");
    std::string s(pad4(n), '\0');
    need(fread(&s[0], 1, s.size(), f_) == s.size(), "ends inside its header");
    s.resize(n);
    return s;
  }
  void skip_atts() {
    const uint32_t tag = get32(), n = get32();
    need(tag == NC_ATTRIBUTE || (tag == 0 && n == 0), "has a damaged attribute list");
    for (uint32_t i = 0; i < n; ++i) {
      getname();
      const int t = (int)get32();
      const uint32_t ne = get32();
      need(fseeko(f_, (off_t)pad4((size_t)ne * tsize(t)), SEEK_CUR) == 0, "ends inside its header");
    }
  }
  std::string path_;
  FILE *f_ = nullptr;
  uint32_t numrecs_ = 0;
  uint64_t recsize_ = 0, last_rec_bytes_ = 0;
  int nrecvars_ = 0;
  std::map<std::string, RVar> vars_;
};

bool file_exists(const std::string &p) { struct stat st; return stat(p.c_str(), &st) == 0; }

// ------------------------------------------------------------------------------------------------ the model's files
struct TracerFile { std::string name, grid, atm, spec; };       // file name; state names of the dynamics' levels, atmosphere_mod's copy, the coefficients
std::vector<TracerFile> tracer_files(const isca_dyn *h, const char *names) {
  std::vector<TracerFile> out;
  if (!h->tracer_on) return out;
  std::vector<std::string> nm;
  if (names && *names) {
    std::string s(names), cur;
    for (char c : s) { if (c == ',') { nm.push_back(cur); cur.clear(); } else if (c != ' ') cur += c; }
    nm.push_back(cur);
  }
  const int nt = std::max(h->cfg.num_tracers, 1);
  for (int k = 0; k < nt; ++k) {
    TracerFile t;
    t.name = (k < (int)nm.size() && !nm[k].empty()) ? nm[k] : (k == 0 ? std::string("sphum") : "tracer" + std::to_string(k + 1));
    const std::string sfx = k == 0 ? "" : std::to_string(k + 1);
    t.grid = "tr" + sfx; t.atm = "tr_atm" + sfx;
    if (k > 0 && h->cfg.tracer_spectral[k]) t.spec = "trs" + sfx;
    out.push_back(t);
  }
  return out;
}
void get(isca_dyn_t *h, const std::string &name, int tl, double *out, size_t n) {
  if (isca_dyn_get_state(h, name.c_str(), tl, out, n)) fail(std::string("spectral_dynamics_end: ") + isca_last_error());
}
void set(isca_dyn_t *h, const std::string &name, int tl, const double *v, size_t n) {
  if (isca_dyn_set_state(h, name.c_str(), tl, v, n)) fail(std::string("spectral_dynamics_init: ") + isca_last_error());
}

// With more than one rank every rank writes and reads files of its own, named as fms_io names the pieces of a distributed file (<name>.nc.NNNN,
// NNNN = the rank): its latitude band of the grid fields, and the spectral arrays at full size with the coefficients of ITS zonal wavenumbers (zero
// elsewhere: get_state / set_state move exactly those).  Such a set is read back by the same number of ranks.
std::string rank_suffix(const isca_dyn *h) {
  if (h->cfg.world_size == 1) return "";
  char buf[16];
  snprintf(buf, sizeof buf, ".%04d", h->cfg.rank);
  return buf;
}
void write_restart(isca_dyn_t *h, const std::string &dir, const char *names) {
  mkdir(dir.c_str(), 0777);
  const std::string sfx = rank_suffix(h);
  const size_t L = h->g.L, J = h->g.Jl, I = h->g.I, N1 = h->g.N1, M1 = h->g.M1;
  const int prev = h->previous, cur = h->current;
  // record nt of a two-level variable = storage slot nt (Fortran time level nt + 1): the current level sits in record `cur`, the previous one
  // in the other; after a cold start both records hold the same values (spectral_dynamics.F90:617-625)
  auto tl_of = [prev, cur](int rec) { return (rec == cur || prev == cur) ? 1 : 0; };
  const auto tracers = tracer_files(h, names);
  auto grid = [&](RestartFile &f, const std::string &fname, const std::string &state, int nrec, size_t z) {
    f.put(fname, nrec, z, J, I, [=](int r, double *out) { get(h, state, nrec == 2 ? tl_of(r) : 1, out, z * J * I); });
  };
  // complex (lev, n, m) state as two real variables; the interleaved values are fetched once per (state, record)
  struct Cache { std::string key; std::vector<double> z; };
  auto cache = std::make_shared<Cache>();
  auto spec = [&](RestartFile &f, const std::string &fname, const std::string &state, size_t z) {
    for (int part = 0; part < 2; ++part)
      f.put(fname + (part ? "_imag" : "_real"), 2, z, N1, M1, [=](int r, double *out) {
        const std::string key = state + "#" + std::to_string(r);
        const size_t n = z * N1 * M1;
        if (cache->key != key) { cache->z.resize(2 * n); get(h, state, tl_of(r), cache->z.data(), 2 * n); cache->key = key; }
        for (size_t i = 0; i < n; ++i) out[i] = cache->z[2 * i + part];
      });
  };
  {
    RestartFile f;       // spectral_dynamics.F90:1502-1531
    f.put("previous", 2, 1, 1, 1, [prev](int, double *out) { out[0] = prev + 1.0; });
    f.put("current", 2, 1, 1, 1, [cur](int, double *out) { out[0] = cur + 1.0; });
    f.put("pk", 2, 1, 1, L + 1, [h, L](int, double *out) { for (size_t k = 0; k <= L; ++k) out[k] = h->tab.pk[k]; });
    f.put("bk", 2, 1, 1, L + 1, [h, L](int, double *out) { for (size_t k = 0; k <= L; ++k) out[k] = h->tab.bk[k]; });
    spec(f, "vors", "vors", L); spec(f, "divs", "divs", L); spec(f, "ts", "ts", L); spec(f, "ln_ps", "ln_ps", 1);
    grid(f, "ug", "ug", 2, L); grid(f, "vg", "vg", 2, L); grid(f, "tg", "tg", 2, L); grid(f, "psg", "psg", 2, 1);
    for (const auto &t : tracers) {
      grid(f, t.name, t.grid, 2, L);
      if (!t.spec.empty()) spec(f, t.name, t.spec, L);
    }
    grid(f, "vorg", "vorg", 1, L); grid(f, "divg", "divg", 1, L); grid(f, "surf_geopotential", "surf_geopotential", 1, 1);
    if (h->cfg.world_size > 1) {     // a piece also carries what one rank re-derives from the spectral state (isca_dyn_refresh_derived): the gradients of T and ln p_s the step keeps
      grid(f, "dxT", "dxT", 1, L); grid(f, "dyT", "dyT", 1, L); grid(f, "dxlp", "dxlp", 1, 1); grid(f, "dylp", "dylp", 1, 1);
    }
    f.close(dir + "/spectral_dynamics.res.nc" + sfx);
  }
  {
    RestartFile f;       // atmosphere.F90:362-375
    f.put("time_pointers", 2, 1, 1, 2, [prev, cur](int, double *out) { out[0] = prev + 1.0; out[1] = cur + 1.0; });
    grid(f, "ug", "ug", 2, L); grid(f, "vg", "vg", 2, L); grid(f, "tg", "tg", 2, L); grid(f, "psg", "psg", 2, 1);
    for (const auto &t : tracers) grid(f, t.name, t.atm, 2, L);
    grid(f, "wg_full", "wg_full", 1, L);
    f.close(dir + "/atmosphere.res.nc" + sfx);
  }
  if (h->cfg.physics == 1) {
    RestartFile f;       // mixed_layer_end
    grid(f, "t_surf", "t_surf", 1, 1);
    f.close(dir + "/mixed_layer.res.nc" + sfx);
  }
}

void read_restart(isca_dyn_t *h, const std::string &dir, const char *names) {
  const std::string sfx = rank_suffix(h);
  const size_t L = h->g.L, J = h->g.Jl, I = h->g.I, N1 = h->g.N1, M1 = h->g.M1;
  Nc3File sd(dir + "/spectral_dynamics.res.nc" + sfx);
  std::unique_ptr<Nc3File> at;
  if (file_exists(dir + "/atmosphere.res.nc" + sfx)) at.reset(new Nc3File(dir + "/atmosphere.res.nc" + sfx));
  auto last = [](const RVar &v, int back) { return v.shape.size() > (size_t)back ? v.shape[v.shape.size() - 1 - back] : (size_t)1; };
  {   // field_size checks of spectral_dynamics.F90:519-545
    const RVar &v = sd.var("vors_real");
    if (last(v, 0) != M1 || last(v, 1) != N1 || last(v, 2) != L)
      fail("spectral_dynamics_init: Resolution of restart data does not match resolution specified on namelist. Restart data: num_fourier=" +
           std::to_string(last(v, 0) - 1) + ", num_spherical=" + std::to_string(last(v, 1) - 1) + ", num_levels=" + std::to_string(last(v, 2)) +
           "  Namelist: num_fourier=" + std::to_string(M1 - 1) + ", num_spherical=" + std::to_string(N1 - 1) + ", num_levels=" + std::to_string(L));
    const RVar &u = sd.var("ug");
    if (last(u, 0) != I || last(u, 1) != J)
      fail("spectral_dynamics_init: Resolution of restart data does not match resolution specified on namelist. Restart data: lon_max=" +
           std::to_string(last(u, 0)) + ", lat_max=" + std::to_string(last(u, 1)) + "  Namelist: lon_max=" + std::to_string(I) + ", lat_max=" + std::to_string(J));
    if (at) { const RVar &a = at->var("ug"); if (last(a, 0) != I || last(a, 1) != J) fail("atmosphere_init: Resolution of restart data does not match resolution specified on namelist."); }
  }
  const int prev = (int)(sd.read("previous", 0)[0] + 0.5) - 1, cur = (int)(sd.read("current", 0)[0] + 0.5) - 1;
  if (prev < 0 || prev > 1 || cur < 0 || cur > 1) fail("read_restart: time pointers out of range");
  if (at) {
    const auto tp = at->read("time_pointers", 0);
    if (tp.size() < 2 || (int)tp[0] - 1 != prev || (int)tp[1] - 1 != cur) fail("read_restart: time pointers of atmosphere.res and spectral_dynamics.res differ");
  }
  for (const char *nm : {"pk", "bk"}) {
    const auto v = sd.read(nm, 0);
    const std::vector<double> &t = (nm[0] == 'p') ? h->tab.pk : h->tab.bk;
    if (v.size() != L + 1 || memcmp(v.data(), t.data(), (L + 1) * 8) != 0)
      fail(std::string("read_restart: ") + nm + " of the restart file differs from the vertical coordinate of the namelist");
  }
  {   // spectral_dynamics.F90:575: the restart file's topography, not get_topography's (the library takes the GLOBAL field: the bands of every rank's file)
    std::vector<double> sg;
    for (int q = 0; q < h->cfg.world_size; ++q) {
      char qs[16]; snprintf(qs, sizeof qs, ".%04d", q);
      std::unique_ptr<Nc3File> other;
      if (h->cfg.world_size > 1 && q != h->cfg.rank) other.reset(new Nc3File(dir + "/spectral_dynamics.res.nc" + qs));
      const auto band = (other ? *other : sd).read("surf_geopotential", 0);
      if (band.size() != J * I) fail("read_restart: surf_geopotential of the restart file does not match the namelist's resolution" + std::string(h->cfg.world_size > 1 ? " and number of ranks" : ""));
      sg.insert(sg.end(), band.begin(), band.end());
    }
    if (isca_dyn_set_surf_geopotential(h, sg.data(), sg.size())) fail(std::string("read_restart: surf_geopotential: ") + isca_last_error());
  }
  if (isca_dyn_set_time_pointers(h, prev, cur, prev == cur ? 0 : 1)) fail(isca_last_error());
  const auto tracers = tracer_files(h, names);
  auto cplx = [&](Nc3File &f, const std::string &fname, int rec) {
    const auto re = f.read(fname + "_real", rec), im = f.read(fname + "_imag", rec);
    std::vector<double> z(2 * re.size());
    for (size_t i = 0; i < re.size(); ++i) { z[2 * i] = re[i]; z[2 * i + 1] = im[i]; }
    return z;
  };
  for (int nt = 0; nt < 2; ++nt) {
    const int tl = (nt == prev && prev != cur) ? 0 : 1;
    if (!(prev != cur || nt == cur)) continue;
    for (const char *nm : {"vors", "divs", "ts", "ln_ps"}) { const auto z = cplx(sd, nm, nt); set(h, nm, tl, z.data(), z.size()); }
    for (const char *nm : {"ug", "vg", "tg", "psg"}) {
      const auto v = sd.read(nm, nt);
      if (at) { const auto a = at->read(nm, nt); if (a.size() != v.size() || memcmp(a.data(), v.data(), v.size() * 8) != 0) fail(std::string("read_restart: ") + nm + " of atmosphere.res and spectral_dynamics.res differ"); }
      set(h, nm, tl, v.data(), v.size());
    }
    for (const auto &t : tracers) {
      if (!sd.has(t.name) || (!t.spec.empty() && !sd.has(t.name + "_real"))) fail("read_restart: tracer " + t.name + " not in the restart file");
      const auto v = sd.read(t.name, nt);
      set(h, t.grid, tl, v.data(), v.size());
      const auto a = (at && at->has(t.name)) ? at->read(t.name, nt) : v;
      set(h, t.atm, tl, a.data(), a.size());
      if (!t.spec.empty()) { const auto z = cplx(sd, t.name, nt); set(h, t.spec, tl, z.data(), z.size()); }
    }
  }
  if (at && at->has("wg_full")) { const auto v = at->read("wg_full", 0); set(h, "wg_full", 1, v.data(), v.size()); }
  if (h->cfg.physics == 1 && file_exists(dir + "/mixed_layer.res.nc" + sfx)) {      // mixed_layer_init: the restart file, else the prescribed distribution
    Nc3File ml(dir + "/mixed_layer.res.nc" + sfx);
    const auto v = ml.read("t_surf", 0);
    if (v.size() != J * I) fail("mixed_layer_init: resolution of mixed_layer.res does not match the namelist");
    set(h, "t_surf", 1, v.data(), v.size());
  }
  if (h->cfg.world_size > 1) {
    for (const char *nm : {"dxT", "dyT", "dxlp", "dylp"}) {
      if (!sd.has(nm)) fail(std::string("read_restart: the piece of a distributed restart file lacks ") + nm);
      const auto v = sd.read(nm, 0);
      set(h, nm, 1, v.data(), v.size());
    }
  } else if (isca_dyn_refresh_derived(h)) fail(isca_last_error());
  // vorg, divg are restart variables of the reference too (spectral_dynamics.F90:1518-1519, read back :566-567): with raw_filter_coeff /= 1 they
  // belong to the new level BEFORE the filter's adjustment (:933-934 vs :1031), which the adjusted spectral state cannot give back
  if (sd.has("vorg") && sd.has("divg")) {
    const auto vo = sd.read("vorg", 0), di = sd.read("divg", 0);
    if (vo.size() == L * J * I && di.size() == L * J * I) { set(h, "vorg", 1, vo.data(), vo.size()); set(h, "divg", 1, di.data(), di.size()); }
  }
}

}  // namespace

#define RS_BEGIN try {
#define RS_END } catch (const std::exception &e) { isca_internal_set_error(e.what()); return 1; } return 0;

extern "C" int isca_dyn_write_restart(isca_dyn_t *h, const char *directory, const char *tracer_names) {
  RS_BEGIN
  if (!h || !directory) fail("null argument");
  write_restart(h, directory, tracer_names);
  RS_END
}
extern "C" int isca_dyn_read_restart(isca_dyn_t *h, const char *directory, const char *tracer_names) {
  RS_BEGIN
  if (!h || !directory) fail("null argument");
  read_restart(h, directory, tracer_names);
  RS_END
}
extern "C" int isca_dyn_restart_exists(const char *directory) {      // (one file, or the pieces of a distributed one)
  return directory && (file_exists(std::string(directory) + "/spectral_dynamics.res.nc") || file_exists(std::string(directory) + "/spectral_dynamics.res.nc.0000")) ? 1 : 0;
}
// A variable of a netCDF classic file for a host that has no netCDF of its own (the Fortran drop-in: INPUT/<topog_file_name>'s zsurf and land_mask, what
// read_data does in get_topography, spectral_init_cond.F90:186-222): record `record` of a record variable (or the whole fixed variable) as doubles.
// count_out = the number of values; they are copied when out != NULL and count >= count_out.
extern "C" int isca_nc_read_variable(const char *path, const char *var_name, int record, double *out, size_t count, size_t *count_out) {
  RS_BEGIN
  if (!path || !var_name) fail("null argument");
  Nc3File f(path);
  const auto v = f.read(var_name, record);
  if (count_out) *count_out = v.size();
  if (out) {
    if (count < v.size()) fail(std::string("read_data: ") + var_name + " of " + path + " holds " + std::to_string(v.size()) + " values, the buffer " + std::to_string(count));
    std::copy(v.begin(), v.end(), out);
  }
  RS_END
}
// The file layer alone, without a device (the CPU tests): writes a small fms_io-style file with known contents to `out_path` (when given) and
// returns in sums[0..2] the sum, the first and the last value of variable `var_name`, record `record`, of `in_path` (when given).
extern "C" int isca_restart_file_selftest(const char *out_path, const char *in_path, const char *var_name, int record, double *sums) {
  RS_BEGIN
  if (out_path && *out_path) {
    RestartFile f;
    f.put("two_level", 2, 3, 4, 5, [](int r, double *out) { for (int i = 0; i < 60; ++i) out[i] = 1000.0 * (r + 1) + i + 0.25; });
    f.put("one_level", 1, 1, 4, 5, [](int, double *out) { for (int i = 0; i < 20; ++i) out[i] = -1.0 / (i + 1); });
    f.put("scalar", 2, 1, 1, 1, [](int r, double *out) { out[0] = r + 1.0; });
    f.close(out_path);
  }
  if (in_path && *in_path) {
    if (!var_name || !sums) fail("null argument");
    Nc3File f(in_path);
    const auto v = f.read(var_name, record);
    double s = 0.0;
    for (double x : v) s += x;
    sums[0] = s; sums[1] = v.empty() ? 0.0 : v.front(); sums[2] = v.empty() ? 0.0 : v.back();
  }
  RS_END
}
