// ephem_file.hip — host-side reader of a legacy JPL DE binary ("Horizons" little-endian file,
// e.g. linux_p1550p2650.440) into the table the kernels consume (outfit_ephem_t).
// SURVEY §8f row n3. Replaces HorizonData::read_horizon_file
// (src/jpl_ephem/horizon/horizon_data.rs:598-684) with parse_ipt :284-305, read_ipt_13_14 :382-405,
// compute_recsize :239-251, parse_all_blocks :527-565 and extract_body_records /
// HorizonRecord::new (:468-491, horizon_records.rs:132-171) for the three bodies that
// HorizonData::ephemeris(Earth, Sun) reads (:810-849): IPT rows 2 (EMB), 9 (Moon), 10 (Sun).
// No device code: pure file parsing, usable without a GPU.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <vector>

#include "../../include/outfit_hip.h"

namespace outfit {
int set_last_error(int code, const std::string& msg);  // api.hip
}

struct outfit_ephem_file {
  outfit_de_header_t header{};
  std::vector<double> coef;  // n_blocks * sum_b 3*ncoef[b]*nsub[b], the packing of outfit_ephem_t
  std::string path;          // for outfit_hip_ephem_body_record (any IPT row, read on demand)
};

namespace {

uint32_t le_u32(const unsigned char* p) {
  return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
double le_f64(const unsigned char* p) {
  uint64_t v = 0;
  for (int i = 7; i >= 0; --i) v = (v << 8) | p[i];
  double d;
  std::memcpy(&d, &v, sizeof d);
  return d;
}
// horizon_data.rs:204-213
unsigned dimension(unsigned i) { return i <= 10 ? 3u : (i == 11 ? 2u : (i <= 13 ? 3u : (i == 14 ? 1u : 0u))); }

constexpr long kOffSS = 252 + 2400;        // after TTL (14*3 char6) and CNAM (400 char6): horizon_data.rs:612-637
constexpr long kOffIpt = 2696;             // :646
constexpr long kOff400thName = 2856;       // :391
constexpr int kBodies[3] = {2, 9, 10};     // EMB, Moon, Sun (horizon_ids.rs / horizon_data.rs:810-849)

}  // namespace

// nothing unwinds through the C ABI (api.hip OUTFIT_ABI_CATCH): allocation failures come back as a code
#define OUTFIT_EPHEM_CATCH                                                                                             \
  catch (const std::bad_alloc&) { return outfit::set_last_error(OUTFIT_ERR_NO_MEMORY, "host allocation failed"); }     \
  catch (...) { return outfit::set_last_error(OUTFIT_ERR_NO_MEMORY, "internal error in the ephemeris reader"); }

extern "C" {

int outfit_hip_ephem_open(const char* path, outfit_ephem_file_t** out) try {
  if (!out) return outfit::set_last_error(OUTFIT_ERR_BAD_HANDLE, "out is NULL");
  *out = nullptr;
  if (!path) return outfit::set_last_error(OUTFIT_ERR_INVALID_BATCH, "path is NULL");
  // the file and the object under construction are owned by holders: every return and every exception releases them
  struct Closer { void operator()(FILE* p) const { if (p) std::fclose(p); } };
  std::unique_ptr<FILE, Closer> file(std::fopen(path, "rb"));
  FILE* f = file.get();
  if (!f) return outfit::set_last_error(OUTFIT_ERR_EPHEM_FILE, std::string("cannot open ") + path);
  auto fail = [&](const std::string& m) {
    return outfit::set_last_error(OUTFIT_ERR_EPHEM_FILE, std::string(path) + ": " + m);
  };
  unsigned char head[4096];  // the reference reads 1 << 12 bytes up front (:627-629)
  if (std::fread(head, 1, sizeof head, f) != sizeof head) return fail("shorter than the 4096-byte header");
  outfit_de_header_t h{};
  h.start_period = le_f64(head + kOffSS);
  h.end_period = le_f64(head + kOffSS + 8);
  h.period_length = le_f64(head + kOffSS + 16);
  h.ncon = (int32_t)le_u32(head + kOffSS + 24);
  h.au_km = le_f64(head + kOffSS + 28);                    // skipped by the reference (:641-642)
  h.earth_moon_mass_ratio = le_f64(head + kOffSS + 36);
  for (int i = 0; i < 36; ++i) h.ipt[i / 3][i % 3] = le_u32(head + kOffIpt + 4 * i);  // parse_ipt :284-305
  h.numde = le_u32(head + kOffIpt + 144);
  for (int c = 0; c < 3; ++c) h.ipt[12][c] = le_u32(head + kOffIpt + 148 + 4 * c);    // LPT
  // the header is untrusted input: bound everything that feeds an offset or a size before using it
  if (h.ncon < 0 || h.ncon > 10000) return fail("implausible NCON (not a little-endian DE file?)");
  // IPT rows 13/14 follow the constant names beyond the 400th (read_ipt_13_14 :382-405). The reference keys
  // this on the DE version named by the caller (>= DE440); the file's own NUMDE is used here.
  if (h.numde >= 440 && h.ncon > 400) {
    const long off = kOff400thName + ((long)h.ncon - 400) * 6;
    unsigned char extra[24];
    if (std::fseek(f, off, SEEK_SET) != 0 || std::fread(extra, 1, 24, f) != 24) return fail("cannot read IPT rows 13/14");
    for (int i = 0; i < 6; ++i) h.ipt[13 + i / 3][i % 3] = le_u32(extra + 4 * i);
  }
  for (unsigned i = 0; i < 15; ++i)
    if (h.ipt[i][1] > 32 || h.ipt[i][2] > 1024) return fail("implausible IPT row (coefficients > 32 or sub-intervals > 1024)");
  uint64_t kernel_size = 4;  // compute_recsize :239-251
  for (unsigned i = 0; i < 15; ++i) kernel_size += 2ull * h.ipt[i][1] * h.ipt[i][2] * dimension(i);
  h.recsize = kernel_size * 4;
  if (h.recsize < 16 || h.recsize > (1ull << 24)) return fail("implausible record size (not a little-endian DE file?)");
  const uint64_t ncoeff = h.recsize / 8;
  uint64_t per_block = 0;  // 64-bit from values bounded above: no wrap
  for (int b = 0; b < 3; ++b) {
    const uint32_t* r = h.ipt[kBodies[b]];
    if (r[0] < 3 || r[1] < 2 || r[2] == 0 || (uint64_t)r[0] - 1 + 3ull * r[1] * r[2] > ncoeff)
      return fail("IPT row of EMB/Moon/Sun points outside the record");
    per_block += 3ull * r[1] * r[2];
  }
  if (std::fseek(f, 0, SEEK_END) != 0) return fail("seek failed");
  const long size = std::ftell(f);
  if (size < (long)(2 * h.recsize)) return fail("no data records");
  h.n_blocks = ((uint64_t)size - 2 * h.recsize) / h.recsize;  // parse_all_blocks :527-565
  if (h.n_blocks == 0) return fail("no data records");
  std::unique_ptr<outfit_ephem_file> holder(new outfit_ephem_file());
  outfit_ephem_file* e = holder.get();
  e->header = h;
  e->path = path;
  e->coef.resize((size_t)h.n_blocks * per_block);
  std::vector<unsigned char> rec((size_t)h.recsize);
  if (std::fseek(f, (long)(2 * h.recsize), SEEK_SET) != 0) return fail("seek failed");
  for (uint64_t nb = 0; nb < h.n_blocks; ++nb) {
    if (std::fread(rec.data(), 1, rec.size(), f) != rec.size()) return fail("truncated data record");
    double* dst = e->coef.data() + (size_t)nb * per_block;
    for (int b = 0; b < 3; ++b) {
      // HorizonRecord::new (horizon_records.rs:132-171): coefficient j of axis a of sub-interval s is
      // block[2 + (offset - 3 + s*ncoef*3) + a*ncoef + j] = block[offset - 1 + (3 s + a) ncoef + j]:
      // the body's segment is contiguous and already in outfit_ephem_t's order.
      const uint32_t* r = h.ipt[kBodies[b]];
      const size_t n = (size_t)(3ull * r[1] * r[2]);
      const unsigned char* src = rec.data() + 8 * ((size_t)r[0] - 1);
      for (size_t k = 0; k < n; ++k) dst[k] = le_f64(src + 8 * k);
      dst += n;
    }
  }
  *out = holder.release();
  return OUTFIT_OK;
} OUTFIT_EPHEM_CATCH

int outfit_hip_ephem_header(const outfit_ephem_file_t* e, outfit_de_header_t* header) {
  if (!e || !header) return outfit::set_last_error(OUTFIT_ERR_BAD_HANDLE, "handle or header is NULL");
  *header = e->header;
  return OUTFIT_OK;
}

int outfit_hip_ephem_view(const outfit_ephem_file_t* e, outfit_ephem_t* view) {
  if (!e || !view) return outfit::set_last_error(OUTFIT_ERR_BAD_HANDLE, "handle or view is NULL");
  const outfit_de_header_t& h = e->header;
  view->jd_start = h.start_period;
  view->jd_end = h.end_period;
  view->block_days = h.period_length;
  view->emrat = h.earth_moon_mass_ratio;
  for (int b = 0; b < 3; ++b) {
    view->ncoef[b] = h.ipt[kBodies[b]][1];
    view->nsub[b] = h.ipt[kBodies[b]][2];
  }
  view->n_blocks = h.n_blocks;
  view->coef = e->coef.data();
  return OUTFIT_OK;
}

// extract_body_records (horizon_data.rs:468-491) for ANY IPT row and one data block, read from the file on demand:
// what `HorizonData.records[block][body]` holds — n_sub HorizonRecords of 3 x n_coeffs coefficients (x, y, z), in
// that order — plus the block's [start_jd, end_jd]. Lets the reference's record known answers for bodies the path
// does not consume (Mercury, Jupiter, Mars: horizon_data.rs:898-1003,1101-1140) be checked through this reader.
int outfit_hip_ephem_body_record(const outfit_ephem_file_t* e, uint32_t ipt_row, uint64_t block, double* jd_range,
                                 double* out, uint64_t capacity) try {
  if (!e || !out) return outfit::set_last_error(OUTFIT_ERR_BAD_HANDLE, "handle or out is NULL");
  const outfit_de_header_t& h = e->header;
  if (ipt_row >= 15 || block >= h.n_blocks) return outfit::set_last_error(OUTFIT_ERR_INVALID_BATCH, "IPT row or block out of range");
  const uint32_t* r = h.ipt[ipt_row];
  const uint64_t n = (uint64_t)dimension(ipt_row) * r[1] * r[2];
  if (n == 0) return outfit::set_last_error(OUTFIT_ERR_INVALID_BATCH, "this IPT row holds no coefficients");
  if (capacity < n) return outfit::set_last_error(OUTFIT_ERR_INVALID_BATCH, "out is too small for this body's record");
  if (r[0] < 3 || (uint64_t)r[0] - 1 + n > h.recsize / 8)
    return outfit::set_last_error(OUTFIT_ERR_EPHEM_FILE, "IPT row points outside the record");
  std::vector<unsigned char> rec((size_t)h.recsize);  // (allocated before the file is opened: nothing to release if it throws)
  FILE* f = std::fopen(e->path.c_str(), "rb");
  if (!f) return outfit::set_last_error(OUTFIT_ERR_EPHEM_FILE, "cannot reopen " + e->path);
  const bool ok = std::fseek(f, (long)((2 + block) * h.recsize), SEEK_SET) == 0 &&
                  std::fread(rec.data(), 1, rec.size(), f) == rec.size();
  std::fclose(f);
  if (!ok) return outfit::set_last_error(OUTFIT_ERR_EPHEM_FILE, e->path + ": truncated data record");
  if (jd_range) { jd_range[0] = le_f64(rec.data()); jd_range[1] = le_f64(rec.data() + 8); }
  const unsigned char* src = rec.data() + 8 * ((size_t)r[0] - 1);
  for (uint64_t k = 0; k < n; ++k) out[k] = le_f64(src + 8 * k);
  return (int)n;
} OUTFIT_EPHEM_CATCH

void outfit_hip_ephem_close(outfit_ephem_file_t* e) { delete e; }

}  // extern "C"
