// see ptau.hpp
#include "ptau.hpp"

#include <cstdio>
#include <cstring>

namespace keaki {
namespace ptau {

std::string SetupFileError::to_string() const {
  char buf[256];
  switch (kind) {
    case ElementSizeMismatch: snprintf(buf, sizeof buf, "Element size mismatch. Obtained: %llu, Expected: %llu", (unsigned long long)a, (unsigned long long)b); break;
    case EmptySection: snprintf(buf, sizeof buf, "Section is uninitialized: %llu", (unsigned long long)a); break;
    case FileError: snprintf(buf, sizeof buf, "File error: \"%s\"", text.c_str()); break;
    case InvalidFileType: snprintf(buf, sizeof buf, "Invalid file type: [%llu, %llu, %llu, %llu]", (unsigned long long)(a & 255), (unsigned long long)((a >> 8) & 255),
                                   (unsigned long long)((a >> 16) & 255), (unsigned long long)((a >> 24) & 255)); break;
    case InvalidNumberOfSections: snprintf(buf, sizeof buf, "Invalid number of sections: %llu", (unsigned long long)a); break;
    case ParseError: snprintf(buf, sizeof buf, "IO error: %s", text.c_str()); break;
    case UnknownSection: snprintf(buf, sizeof buf, "Unknown section ID: %llu", (unsigned long long)a); break;
    case OffCurve: snprintf(buf, sizeof buf, "%llu point(s) of section %s are not on the curve (first at index %llu)", (unsigned long long)a, text.c_str(), (unsigned long long)b); break;
    case Truncated: snprintf(buf, sizeof buf, "File is truncated: %llu bytes needed at offset %llu", (unsigned long long)a, (unsigned long long)b); break;
  }
  return buf;
}

static SetupFileError err(SetupFileError::Kind k, uint64_t a = 0, uint64_t b = 0, std::string t = "") { return SetupFileError{k, a, b, std::move(t)}; }
static uint32_t le32(const uint8_t* p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24; }
static uint64_t le64(const uint8_t* p) { return (uint64_t)le32(p) | (uint64_t)le32(p + 4) << 32; }

int section_index(uint8_t id) {
  switch (id) {
    case 1: return 0; case 2: return 1; case 3: return 2; case 4: return 3; case 5: return 4; case 6: return 5; case 7: return 6;
    case 12: return 7; case 13: return 8; case 14: return 9; case 15: return 10;
    default: return -1;
  }
}

FileResult<std::vector<uint8_t>> load(const std::string& path) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return FileResult<std::vector<uint8_t>>::Err(err(SetupFileError::FileError, 0, 0, std::string("cannot open ") + path));
  std::vector<uint8_t> data;
  uint8_t buf[1 << 16];
  size_t got;
  while ((got = fread(buf, 1, sizeof buf, f)) > 0) data.insert(data.end(), buf, buf + got);
  bool bad = ferror(f) != 0;
  fclose(f);
  if (bad) return FileResult<std::vector<uint8_t>>::Err(err(SetupFileError::FileError, 0, 0, std::string("read error on ") + path));
  return FileResult<std::vector<uint8_t>>::Ok(std::move(data));
}

FileResult<bool> verify_metadata(const std::vector<uint8_t>& d) {
  if (d.size() < METADATA_LEN) return FileResult<bool>::Err(err(SetupFileError::Truncated, METADATA_LEN, 0));
  if (memcmp(d.data(), "ptau", 4) != 0) return FileResult<bool>::Err(err(SetupFileError::InvalidFileType, le32(d.data())));
  uint32_t n = le32(d.data() + 8);   // bytes 4..8 hold the version
  if (n != N_SECTIONS) return FileResult<bool>::Err(err(SetupFileError::InvalidNumberOfSections, n));
  return FileResult<bool>::Ok(true);
}

FileResult<SectionInfo> section_info_from_data(const uint8_t h[SECTION_HEADER_LEN], size_t offset) {
  if (section_index(h[0]) < 0) return FileResult<SectionInfo>::Err(err(SetupFileError::UnknownSection, h[0]));  // one byte of the id is enough (:174)
  SectionInfo s;
  s.id = h[0];
  s.size = le64(h + 4);
  s.position = offset + SECTION_HEADER_LEN;
  return FileResult<SectionInfo>::Ok(s);
}

FileResult<FileSections> parse_sections(const std::vector<uint8_t>& d) {
  FileSections fs;
  size_t offset = METADATA_LEN;
  for (size_t i = 0; i < N_SECTIONS; i++) {
    if (offset + SECTION_HEADER_LEN > d.size()) return FileResult<FileSections>::Err(err(SetupFileError::Truncated, SECTION_HEADER_LEN, offset));
    auto si = section_info_from_data(d.data() + offset, offset);
    if (!si.ok) return FileResult<FileSections>::Err(si.error);
    if (si.value.size > d.size() || si.value.position + si.value.size > d.size())
      return FileResult<FileSections>::Err(err(SetupFileError::Truncated, si.value.size, si.value.position));
    fs.sections[i] = si.value;     // slot i = i-th section of the file, as in the reference (:134-141)
    offset += SECTION_HEADER_LEN + (size_t)si.value.size;
  }
  return FileResult<FileSections>::Ok(fs);
}

static const SectionInfo& get(const FileSections& s, uint8_t id) { return s.sections[(size_t)section_index(id)]; }

FileResult<HeaderSection> parse_header(const std::vector<uint8_t>& d, const FileSections& s) {
  const SectionInfo& si = get(s, 1);
  size_t off = si.position;
  if (off + 4 > d.size()) return FileResult<HeaderSection>::Err(err(SetupFileError::Truncated, 4, off));
  uint32_t n8 = le32(d.data() + off);
  off += 4;
  if (off + (size_t)n8 + 8 > d.size()) return FileResult<HeaderSection>::Err(err(SetupFileError::Truncated, (uint64_t)n8 + 8, off));
  HeaderSection h;
  h.field_modulus.assign(d.begin() + off, d.begin() + off + n8);
  off += n8;
  h.power = le32(d.data() + off);
  h.ceremony_power = le32(d.data() + off + 4);
  return FileResult<HeaderSection>::Ok(std::move(h));
}

template <class Pt, size_t BYTES>
static FileResult<std::vector<Pt>> parse_points(const std::vector<uint8_t>& d, const SectionInfo& si, uint64_t n_elements) {
  if (si.size != BYTES * n_elements) return FileResult<std::vector<Pt>>::Err(err(SetupFileError::ElementSizeMismatch, BYTES * n_elements, si.size));
  std::vector<Pt> out((size_t)n_elements);
  for (size_t i = 0; i < (size_t)n_elements; i++) memcpy(out[i].w.data(), d.data() + si.position + BYTES * i, BYTES);  // little-endian limbs, as stored
  return FileResult<std::vector<Pt>>::Ok(std::move(out));
}
FileResult<std::vector<G1>> parse_tau_g1(const std::vector<uint8_t>& d, const FileSections& s, uint32_t power) {
  if (power > 31) return FileResult<std::vector<G1>>::Err(err(SetupFileError::ParseError, 0, 0, "power out of range"));
  return parse_points<G1, 64>(d, get(s, 2), ((uint64_t)1 << power) * 2 - 1);
}
FileResult<std::vector<G2>> parse_tau_g2(const std::vector<uint8_t>& d, const FileSections& s, uint32_t power) {
  if (power > 31) return FileResult<std::vector<G2>>::Err(err(SetupFileError::ParseError, 0, 0, "power out of range"));
  return parse_points<G2, 128>(d, get(s, 3), (uint64_t)1 << power);
}

FileResult<PowersOfTau> get_powers_from_file(const std::string& path) {
  typedef FileResult<PowersOfTau> R;
  auto data = load(path);
  if (!data.ok) return R::Err(data.error);
  auto md = verify_metadata(data.value);
  if (!md.ok) return R::Err(md.error);
  auto secs = parse_sections(data.value);
  if (!secs.ok) return R::Err(secs.error);
  auto hdr = parse_header(data.value, secs.value);
  if (!hdr.ok) return R::Err(hdr.error);
  auto g1 = parse_tau_g1(data.value, secs.value, hdr.value.power);
  if (!g1.ok) return R::Err(g1.error);
  auto g2 = parse_tau_g2(data.value, secs.value, hdr.value.power);
  if (!g2.ok) return R::Err(g2.error);
  return R::Ok(PowersOfTau{std::move(g1.value), std::move(g2.value), std::move(hdr.value)});
}

}  // namespace ptau

#ifndef KEAKI_PTAU_PARSE_ONLY          /* the sanitizer harness (ptau_fuzz_main.cpp) builds the parser alone: no GPU library behind it */
namespace kzg {
FileSetup new_from_file(std::shared_ptr<Device> dev, const std::string& path) {
  auto fail = [](ptau::SetupFileError e) { return FileSetup{false, nullptr, std::move(e)}; };
  auto pw = ptau::get_powers_from_file(path);
  if (!pw.ok) return fail(pw.error);
  if (pw.value.g2.size() < 2)     // g2_aff.get(1) ... ok_or(EmptySection(3))  (src/kzg.rs:35-39)
    return fail(ptau::SetupFileError{ptau::SetupFileError::EmptySection, 3, 0, ""});
  uint64_t bad = 0, first = 0;
  dev->check(keaki_hip_g2_check(dev->ctx(), pw.value.g2[0].w.data(), pw.value.g2.size(), &bad, &first));
  if (bad) return fail(ptau::SetupFileError{ptau::SetupFileError::OffCurve, bad, first, "3 (tauG2)"});
  auto s = std::make_unique<KZGSetup>(KZGSetup::from_powers(dev, std::move(pw.value.g1), pw.value.g2[1]));
  dev->check(keaki_hip_srs_g1_check(dev->ctx(), s->srs(), &bad, &first));
  if (bad) return fail(ptau::SetupFileError{ptau::SetupFileError::OffCurve, bad, first, "2 (tauG1)"});
  return FileSetup{true, std::move(s), ptau::SetupFileError{ptau::SetupFileError::FileError, 0, 0, ""}};
}
}  // namespace kzg
#endif
}  // namespace keaki
