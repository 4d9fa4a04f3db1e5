// History files from the library itself: diag_manager's part for the fields of the hot path.
//
// The reference's spectral_diagnostics (src/atmos_spectral/model/spectral_dynamics.F90:1705-1867) hands 20 dynamics fields of the new time level to
// diag_manager with send_data at the end of every atmosphere call (the registrations: :1554-1700); diag_manager reads the run directory's
// `diag_table`, averages each field over its file's output interval and writes one record per interval (src/shared/diag_manager).  Here the sums are
// accumulated ON THE DEVICE by the step itself (k_diag_accumulate; isca_dyn_diag_select / isca_dyn_diag_read), and this file is the rest of it for a host
// that has no Python around it -- the Fortran drop-in (bindings/fortran/dropin: spectral_dynamics_init opens, spectral_dynamics_end closes):
//   isca_dyn_diag_open   parses the diag_table (title, base date, file lines, field lines -- the format of src/extra/python/isca/diagtable.py and of
//                        every exp/test_cases script), selects the union of the fields on the device and creates <directory>/<file>.nc;
//   every step           (api.hip: isca_dyn_step / isca_dyn_dynamics call isca_history_after_step) counts; at the end of a chunk of steps -- the
//                        greatest common divisor of the files' intervals -- the sums come off the device, are added to each file's own, and a file whose
//                        interval is complete gets its record appended;
//   isca_dyn_diag_close  closes the files.
// Files: netCDF classic (nc3.h) with the reference's names -- lon, lat, pfull, phalf, time, average_T1 / _T2 / _DT, static pk, bk, and the fields as
// (time, pfull, lat, lon) / (time, lat, lon) with long_name, units, cell_methods -- the same files, value for value, as the Python host mirror's
// (isca_amd/diag.py: History), whose arithmetic (device mean x count, sum of the chunks, / steps of the interval) is repeated here in its order.
// With more than one rank every rank writes its latitude band as <file>.nc.NNNN, as diag_manager names the pieces of a distributed file.
// Fields of other modules than `dynamics` (and, with the moist package, atmosphere: precipitation, mixed_layer: t_surf) are not the device core's:
// an entry that asks for one is refused by name -- nothing is dropped silently.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <memory>
#include <numeric>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>
#include <sys/stat.h>
#include "core.h"
#include "nc3.h"

void isca_internal_set_error(const std::string &m);       // api.hip: the thread's isca_last_error() text

namespace {

using isca_nc3::Nc3Writer;
[[noreturn]] void fail(const std::string &m) { throw std::runtime_error(m); }

struct FieldInfo { const char *name, *long_name, *units, *state, *module; bool two_d; };
// name -> (long name, units) as registered by the reference (spectral_dynamics.F90:1604-1690; idealized_moist_phys.F90:672, mixed_layer.F90:359);
// state: the state array an instantaneous (time_avg = .false.) sample is taken from
const FieldInfo FIELDS[] = {
    {"ps", "surface pressure", "pascals", "psg", "dynamics", true},
    {"ucomp", "zonal wind component", "m/sec", "ug", "dynamics", false},
    {"vcomp", "meridional wind component", "m/sec", "vg", "dynamics", false},
    {"temp", "temperature", "deg_k", "tg", "dynamics", false},
    {"vor", "vorticity", "sec**-1", "vorg", "dynamics", false},
    {"div", "divergence", "sec**-1", "divg", "dynamics", false},
    {"omega", "dp/dt vertical velocity", "Pa/sec", "wg_full", "dynamics", false},
    {"sphum", "specific humidity", "kg/kg", "tr", "dynamics", false},
    {"ucomp_sq", "zonal wind squared", "(m/sec)**2", nullptr, "dynamics", false},
    {"vcomp_sq", "meridional wind squared", "(m/sec)**2", nullptr, "dynamics", false},
    {"ucomp_vcomp", "zonal wind * meridional wind", "(m/sec)**2", nullptr, "dynamics", false},
    {"temp_sq", "temperature squared", "deg_k**2", nullptr, "dynamics", false},
    {"ucomp_temp", "zonal wind * temperature", "m*K/sec", nullptr, "dynamics", false},
    {"vcomp_temp", "meridional wind * temperature", "m*K/sec", nullptr, "dynamics", false},
    {"omega_sq", "omega squared", "(Pa/sec)**2", nullptr, "dynamics", false},
    {"omega_temp", "dp/dt * temperature", "Pa*K/sec", nullptr, "dynamics", false},
    {"ucomp_omega", "vertical * zonal wind", "m*Pa/sec**2", nullptr, "dynamics", false},
    {"vcomp_omega", "vertical * meridional wind", "m*Pa/sec**2", nullptr, "dynamics", false},
    {"vcomp_vor", "meridional wind * vorticity", "m/sec**2", nullptr, "dynamics", false},
    {"wspd", "wind speed", "m/sec", nullptr, "dynamics", false},
    {"precipitation", "precipitation from resolved, parameterised and snow", "kg/m/m/s", "precip", "atmosphere", true},
    {"t_surf", "surface temperature", "K", "t_surf", "mixed_layer", true},
};
const FieldInfo *field_info(const std::string &nm) {
  for (const auto &f : FIELDS) if (nm == f.name) return &f;
  return nullptr;
}
struct StaticInfo { const char *name, *long_name, *units; };
const StaticInfo STATICS[] = {{"pk", "vertical coordinate pressure values", "pascals"}, {"bk", "vertical coordinate sigma values", "none"}};

double unit_seconds(const std::string &u, const std::string &what) {
  if (u == "seconds") return 1; if (u == "minutes") return 60; if (u == "hours") return 3600; if (u == "days") return 86400;
  fail("diag_table: unsupported " + what + " '" + u + "' (seconds, minutes, hours or days)");
}

struct HistField {
  const FieldInfo *info;
  std::string out_name;
  bool avg;
  std::vector<double> sum;          // this file's own sum over the chunks of the interval (time_avg fields)
};
struct HistFile {
  std::string name, path, time_units;
  double interval_s = 0, scale = 1;
  long every = 0, nsum = 0, elapsed = 0;
  std::vector<HistField> fields;
  std::vector<const StaticInfo *> statics;
  Nc3Writer w;
  std::vector<std::vector<double>> rec;      // the record being written: one array per field
  double t1 = 0, t2 = 0;
};

}  // namespace

struct isca_history {
  int base_date[6] = {0, 0, 0, 0, 0, 0};     // the table's second line: year month day hour minute second (diag_manager writes it into the time units)
  std::vector<std::unique_ptr<HistFile>> files;
  std::vector<std::string> names;            // union of the files' fields: what the device accumulates
  std::vector<std::vector<double>> chunk;    // the chunk's sums, one per name
  long chunk_steps = 1, since = 0;
  double t0 = 0, dt = 0;
};

namespace {

// ---- the diag_table: comma-separated, strings quoted, '#' comments (diag_manager's parser: diag_table.F90; the harness's writer: diagtable.py:5-35)
std::vector<std::string> split_entry(const std::string &line) {
  std::vector<std::string> out;
  std::string cur;
  bool quoted = false, any = false;
  for (char c : line) {
    if (c == '"' || c == '\'') { quoted = !quoted; any = true; continue; }
    if (c == ',' && !quoted) { out.push_back(cur); cur.clear(); any = false; continue; }
    if (!quoted && (c == ' ' || c == '\t' || c == '\r')) continue;
    cur.push_back(c); any = true;
  }
  if (any || !cur.empty()) out.push_back(cur);
  return out;
}
bool is_number(const std::string &s) {
  if (s.empty()) return false;
  char *e = nullptr;
  strtod(s.c_str(), &e);
  return e && *e == 0;
}
std::string lower(std::string s) { for (auto &c : s) c = (char)tolower(c); return s; }

void parse_table(const isca_dyn *h, const std::string &text, const std::string &dir, isca_history &H) {
  std::istringstream in(text);
  std::string line;
  int header_lines = 0;
  const std::string sfx = [&] { if (h->cfg.world_size == 1) return std::string(); char b[16]; snprintf(b, sizeof b, ".%04d", h->cfg.rank); return std::string(b); }();
  while (std::getline(in, line)) {
    const size_t first = line.find_first_not_of(" \t\r");
    if (first == std::string::npos || line[first] == '#') continue;
    if (header_lines < 2) {                                           // the title, then the base date ("0 0 0 0 0 0" without a calendar, diagtable.py:13-17)
      if (header_lines == 1) {
        std::istringstream bd(line);
        for (int i = 0; i < 6; ++i) { int v = 0; if (bd >> v) H.base_date[i] = v; }
      }
      ++header_lines; continue;
    }
    const std::vector<std::string> t = split_entry(line);
    if (t.size() < 2) continue;
    if (is_number(t[1])) {                                            // "file_name", output_freq, "output_units", format, "time_units", "long_name"
      if (t.size() < 5) fail("diag_table: a file line needs file_name, output_freq, output_units, format, time_units: " + line);
      auto f = std::make_unique<HistFile>();
      f->name = t[0];
      const double freq = atof(t[1].c_str());
      if (!(freq > 0)) fail("diag_table: file '" + t[0] + "': output_freq must be positive (0 / -1, the end-of-run and every-step files, are not supported)");
      f->interval_s = freq * unit_seconds(lower(t[2]), "output_units");
      f->time_units = lower(t[4]);
      f->scale = unit_seconds(f->time_units, "time_units");
      f->path = dir + "/" + t[0] + ".nc" + sfx;
      H.files.push_back(std::move(f));
    } else {                                                          // "module", "field", "output_name", "file", "time_sampling", time_avg, "other_opts", precision
      if (t.size() < 6) fail("diag_table: a field line needs module_name, field_name, output_name, file_name, time_sampling, time_avg: " + line);
      HistFile *file = nullptr;
      for (auto &f : H.files) if (f->name == t[3]) file = f.get();
      if (!file) fail("diag_table: field '" + t[1] + "' names the file '" + t[3] + "', which the table does not define before it");
      const std::string tm = lower(t[5]);
      bool avg;
      if (tm == ".true." || tm == "mean" || tm == "average" || tm == "avg") avg = true;
      else if (tm == ".false." || tm == "none") avg = false;
      else fail("diag_table: field '" + t[1] + "': time_avg '" + t[5] + "' is not a supported value (.true. / .false.)");
      bool is_static = false;
      for (const auto &s : STATICS)
        if (t[1] == s.name && t[0] == "dynamics") { file->statics.push_back(&s); is_static = true; }
      if (is_static) continue;
      const FieldInfo *fi = field_info(t[1]);
      if (!fi || t[0] != fi->module)
        fail("diag_table: field '" + t[1] + "' of module '" + t[0] + "' is not one the device core accumulates (module dynamics: spectral_diagnostics' 20 fields, pk, bk; "
             "with the moist package atmosphere: precipitation and mixed_layer: t_surf)");
      if (fi->two_d && std::string(fi->module) != "dynamics" && h->cfg.physics != 1) fail("diag_table: field '" + t[1] + "' exists only with the moist physics package");
      if (!avg && !fi->state) fail("diag_table: " + t[1] + " is only available as a time average");
      file->fields.push_back(HistField{fi, t[2].empty() ? t[1] : t[2], avg, {}});
    }
  }
  // files without fields are not written (diag_manager does the same)
  std::vector<std::unique_ptr<HistFile>> keep;
  for (auto &f : H.files) if (!f->fields.empty() || !f->statics.empty()) keep.push_back(std::move(f));
  H.files = std::move(keep);
}

void get_table(isca_dyn *h, const char *nm, std::vector<double> &v, size_t n) {
  v.resize(n);
  if (isca_dyn_get_table(h, nm, v.data(), n)) fail(std::string("diag_manager: ") + isca_last_error());
}

void create_file(isca_dyn *h, const isca_history &H, HistFile &f) {
  const isca::Geom &g = h->g;
  const size_t I = g.I, J = g.Jl, L = g.L;
  Nc3Writer &w = f.w;
  const int dt = w.dim("time", 0);
  std::vector<double> lon, lat_all, pk, bk;
  get_table(h, "deg_lon", lon, g.I); get_table(h, "deg_lat", lat_all, g.J); get_table(h, "pk", pk, L + 1); get_table(h, "bk", bk, L + 1);
  std::vector<double> lat(lat_all.begin() + g.j0, lat_all.begin() + g.j0 + J);
  const int dlon = w.dim("lon", I), dlat = w.dim("lat", J);
  w.var("lon", {dlon}, {{"units", "degrees_E"}, {"cartesian_axis", "X"}}, [lon](int, double *o) { std::copy(lon.begin(), lon.end(), o); });
  w.var("lat", {dlat}, {{"units", "degrees_N"}, {"cartesian_axis", "Y"}}, [lat](int, double *o) { std::copy(lat.begin(), lat.end(), o); });
  // approximate pressure levels of the axes, hPa (spectral_dynamics.F90:1583-1589): the Simmons-Burridge full levels of the reference surface pressure
  std::vector<double> p_half(L + 1), lnp(L + 1), p_full(L);
  for (size_t k = 0; k <= L; ++k) {
    p_half[k] = (pk[k] + bk[k] * h->cfg.reference_sea_level_press) / 100.0;
    lnp[k] = std::log(p_half[k] > 0 ? p_half[k] : 1.0);
  }
  for (size_t k = 0; k < L; ++k) {
    if (p_half[k] == 0.0) p_full[k] = std::exp(lnp[k + 1] - 1.0);
    else p_full[k] = std::exp(lnp[k + 1] - (1.0 - p_half[k] * (lnp[k + 1] - lnp[k]) / (p_half[k + 1] - p_half[k])));
  }
  const int dph = w.dim("phalf", L + 1), dpf = w.dim("pfull", L);
  w.var("phalf", {dph}, {{"units", "hPa"}, {"cartesian_axis", "Z"}, {"positive", "down"}}, [p_half](int, double *o) { std::copy(p_half.begin(), p_half.end(), o); });
  w.var("pfull", {dpf}, {{"units", "hPa"}, {"cartesian_axis", "Z"}, {"positive", "down"}}, [p_full](int, double *o) { std::copy(p_full.begin(), p_full.end(), o); });
  HistFile *fp = &f;
  char since[64];       // diag_util.F90:1582-1585: a, ' since ', i4.4, '-', i2.2, '-', i2.2, ' ', i2.2, ':', i2.2, ':', i2.2
  snprintf(since, sizeof since, " since %04d-%02d-%02d %02d:%02d:%02d", H.base_date[0], H.base_date[1], H.base_date[2], H.base_date[3], H.base_date[4], H.base_date[5]);
  w.var("time", {dt}, {{"units", f.time_units + since}, {"cartesian_axis", "T"}}, [fp](int, double *o) { o[0] = 0.5 * (fp->t1 + fp->t2) / fp->scale; });
  w.var("average_T1", {dt}, {}, [fp](int, double *o) { o[0] = fp->t1 / fp->scale; });
  w.var("average_T2", {dt}, {}, [fp](int, double *o) { o[0] = fp->t2 / fp->scale; });
  w.var("average_DT", {dt}, {}, [fp](int, double *o) { o[0] = (fp->t2 - fp->t1) / fp->scale; });
  for (const StaticInfo *s : f.statics) {
    const std::vector<double> &v = std::string(s->name) == "pk" ? pk : bk;
    w.var(s->name, {dph}, {{"long_name", s->long_name}, {"units", s->units}}, [v](int, double *o) { std::copy(v.begin(), v.end(), o); });
  }
  f.rec.resize(f.fields.size());
  for (size_t i = 0; i < f.fields.size(); ++i) {
    const HistField &fld = f.fields[i];
    std::vector<std::pair<std::string, std::string>> atts = {{"long_name", fld.info->long_name}, {"units", fld.info->units}};
    if (fld.avg) { atts.push_back({"cell_methods", "time: mean"}); atts.push_back({"time_avg_info", "average_T1,average_T2,average_DT"}); }
    const std::vector<int> dims = fld.info->two_d ? std::vector<int>{dt, dlat, dlon} : std::vector<int>{dt, dpf, dlat, dlon};
    w.var(fld.out_name, dims, atts, [fp, i](int, double *o) { std::copy(fp->rec[i].begin(), fp->rec[i].end(), o); });
  }
  w.begin(f.path);
}

size_t field_count(const isca_dyn *h, const FieldInfo *fi) { return (size_t)h->g.Jl * h->g.I * (fi->two_d ? 1 : h->g.L); }

// the interval of file f is complete: its record (History._flush of isca_amd/diag.py)
void flush_file(isca_dyn *h, isca_history &H, HistFile &f) {
  for (size_t i = 0; i < f.fields.size(); ++i) {
    HistField &fld = f.fields[i];
    const size_t n = field_count(h, fld.info);
    f.rec[i].resize(n);
    if (fld.avg) {
      for (size_t k = 0; k < n; ++k) f.rec[i][k] = fld.sum[k] / (double)f.nsum;
      fld.sum.clear();
    } else if (isca_dyn_get_state(h, fld.info->state, 1, f.rec[i].data(), n)) fail(std::string("diag_manager: ") + isca_last_error());      // the sample at the end of the interval
  }
  f.nsum = 0;
  f.t2 = H.t0 + (double)f.elapsed * H.dt;
  f.t1 = f.t2 - f.interval_s;
  f.w.append();
}

// a chunk of steps is complete: the sums off the device (mean x count, as the Python collector forms them), into every file
void take_chunk(isca_dyn *h, isca_history &H) {
  const long nsteps = H.since;
  H.since = 0;
  long cnt = nsteps;
  for (size_t j = 0; j < H.names.size(); ++j) {
    const FieldInfo *fi = field_info(H.names[j]);
    const size_t n = field_count(h, fi);
    H.chunk[j].resize(n);
    if (isca_dyn_diag_read(h, H.names[j].c_str(), H.chunk[j].data(), n, &cnt, 0)) fail(std::string("diag_manager: ") + isca_last_error());
    for (size_t k = 0; k < n; ++k) H.chunk[j][k] = H.chunk[j][k] * (double)cnt;
  }
  if (!H.names.empty()) {
    if (cnt != nsteps) fail("diag_manager: the device accumulated " + std::to_string(cnt) + " steps, " + std::to_string(nsteps) + " expected (isca_dyn_diag_select / _read called beside an open diag_table?)");
    if (isca_dyn_diag_read(h, H.names[0].c_str(), nullptr, 0, &cnt, 1)) fail(std::string("diag_manager: ") + isca_last_error());
  }
  for (auto &fp : H.files) {
    HistFile &f = *fp;
    for (HistField &fld : f.fields) {
      if (!fld.avg) continue;
      size_t j = 0;
      while (H.names[j] != fld.info->name) ++j;
      if (fld.sum.empty()) fld.sum = H.chunk[j];
      else for (size_t k = 0; k < fld.sum.size(); ++k) fld.sum[k] = fld.sum[k] + H.chunk[j][k];
    }
    f.nsum += nsteps; f.elapsed += nsteps;
    if (f.elapsed % f.every == 0) flush_file(h, H, f);
  }
}

}  // namespace

// called by the step loops of api.hip after every completed step
void isca_history_after_step(isca_dyn *h) {
  isca_history &H = *h->hist;
  if (++H.since >= H.chunk_steps) take_chunk(h, H);
}
void isca_history_destroy(isca_dyn *h) {
  delete h->hist;
  h->hist = nullptr;
  h->hist_wg_full = false;
}

#define HS_BEGIN try {
#define HS_END } catch (const std::exception &e) { isca_internal_set_error(e.what()); return 1; } return 0;

extern "C" int isca_dyn_diag_open(isca_dyn_t *h, const char *diag_table, const char *directory, double start_seconds) {
  HS_BEGIN
  if (!h || !diag_table) fail("null argument");
  if (h->hist) fail("diag_manager_init: a diag_table is already open on this handle (isca_dyn_diag_close first)");
  std::string text(diag_table);
  if (text.find('\n') == std::string::npos) {          // a path: the run directory's file
    std::ifstream in(text);
    if (!in) fail("diag_manager_init: cannot open " + text);
    std::stringstream ss; ss << in.rdbuf();
    text = ss.str();
  }
  const std::string dir = (directory && *directory) ? directory : ".";
  auto H = std::make_unique<isca_history>();
  parse_table(h, text, dir, *H);
  H->t0 = start_seconds; H->dt = h->cfg.dt_atmos;
  if (H->files.empty()) return 0;                      // (a table without entries for the device core: nothing to do, nothing open)
  mkdir(dir.c_str(), 0777);
  long g = 0;
  for (auto &f : H->files) {
    const double steps = f->interval_s / H->dt;
    if (std::fabs(steps - std::round(steps)) > 1e-9 || steps < 1) fail("diag_table: the output interval of '" + f->name + "' must be a multiple of dt_atmos");
    f->every = (long)std::llround(steps);
    g = std::gcd(g, f->every);
    for (const HistField &fld : f->fields) {
      bool have = false;
      for (const auto &nm : H->names) have = have || nm == fld.info->name;
      if (!have && fld.avg) H->names.push_back(fld.info->name);
      // an instantaneous sample of omega reads wg_full, which a step only stores on request (isca_dyn_step: store_wg); a record's interval may end on
      // any step of a multi-step call
      if (!fld.avg && fld.info->state && std::strcmp(fld.info->state, "wg_full") == 0) h->hist_wg_full = true;
    }
  }
  H->chunk_steps = g;
  H->chunk.resize(H->names.size());
  std::string csv;
  for (const auto &nm : H->names) csv += (csv.empty() ? "" : ",") + nm;
  if (isca_dyn_diag_select(h, csv.c_str())) fail(std::string("diag_manager_init: ") + isca_last_error());
  for (auto &f : H->files) create_file(h, *H, *f);
  h->hist = H.release();
  HS_END
}

extern "C" int isca_dyn_diag_close(isca_dyn_t *h) {
  HS_BEGIN
  if (!h) fail("null argument");
  if (!h->hist) return 0;
  for (auto &f : h->hist->files) f->w.finish();
  const bool selected = !h->hist->names.empty();
  isca_history_destroy(h);
  if (selected && isca_dyn_diag_select(h, "")) fail(std::string("diag_manager_end: ") + isca_last_error());
  HS_END
}
