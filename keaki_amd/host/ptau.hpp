// snarkjs `.ptau` reader -- host mirror of the reference's src/kzg/ptau.rs (scope row f-3: SRS ingest).
// Same section model, the same checks and the same error variants as the reference; one deliberate difference: the reference
// hands the 64-byte records to `deserialize_uncompressed_unchecked` (src/kzg/ptau.rs:266,314), which takes the bytes as canonical
// integers although snarkjs writes Montgomery residues, and never checks the curve equation. Here the records are kept as the
// Montgomery limbs they are (which is exactly the device layout of include/keaki_hip.h, so they are uploaded verbatim) and
// KZGSetup::new_from_file has the GPU verify y^2 = x^3 + b for every point (SetupFileError::OffCurve).
#pragma once
#include <array>
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "keaki.hpp"

namespace keaki {
namespace ptau {

constexpr size_t N_SECTIONS = 11;         // src/kzg/ptau.rs:14
constexpr size_t METADATA_LEN = 12;       // :16
constexpr size_t SECTION_HEADER_LEN = 12; // :18

// src/kzg/ptau.rs:360-376 (+ OffCurve, Truncated: conditions the reference does not detect / panics on)
struct SetupFileError {
  enum Kind { ElementSizeMismatch, EmptySection, FileError, InvalidFileType, InvalidNumberOfSections, ParseError, UnknownSection,
              OffCurve, Truncated } kind;
  uint64_t a = 0, b = 0;   // the variant's numeric payload ({0}, {1})
  std::string text;        // the variant's string payload
  std::string to_string() const;
};
template <class T>
struct FileResult {
  bool ok;
  T value;
  SetupFileError error;
  static FileResult Ok(T v) { return FileResult{true, std::move(v), SetupFileError{SetupFileError::FileError, 0, 0, ""}}; }
  static FileResult Err(SetupFileError e) { return FileResult{false, T(), std::move(e)}; }
};

// src/kzg/ptau.rs:24-69: section ids and their slot in the file
int section_index(uint8_t id);   // -1 for an unknown id (SetupFileError::UnknownSection)

struct SectionInfo { uint8_t id = 1; uint64_t size = 0; size_t position = 0; };            // :157-183
struct FileSections { std::array<SectionInfo, N_SECTIONS> sections; };                       // :123-153
struct HeaderSection { std::vector<uint8_t> field_modulus; uint32_t power = 0, ceremony_power = 0; };  // :187-226

FileResult<std::vector<uint8_t>> load(const std::string& path);                              // FileLoader::load :107-119
FileResult<bool> verify_metadata(const std::vector<uint8_t>& data);                          // :325-344
FileResult<SectionInfo> section_info_from_data(const uint8_t header[SECTION_HEADER_LEN], size_t offset);  // :169-182
FileResult<FileSections> parse_sections(const std::vector<uint8_t>& data);                   // :129-145
FileResult<HeaderSection> parse_header(const std::vector<uint8_t>& data, const FileSections& s);  // :198-225
// [tau^i]_1, 2 * 2^power - 1 points (:230-275) and [tau^i]_2, 2^power points (:278-322); limbs verbatim (Montgomery)
FileResult<std::vector<G1>> parse_tau_g1(const std::vector<uint8_t>& data, const FileSections& s, uint32_t power);
FileResult<std::vector<G2>> parse_tau_g2(const std::vector<uint8_t>& data, const FileSections& s, uint32_t power);
struct PowersOfTau { std::vector<G1> g1; std::vector<G2> g2; HeaderSection header; };
FileResult<PowersOfTau> get_powers_from_file(const std::string& path);                       // :347-358

}  // namespace ptau

namespace kzg {
// src/kzg.rs:33-52 `KZGSetup::new_from_file`: g1 powers from section 2, tau_g2 = second point of section 3; plus the on-device
// curve check described above.
struct FileSetup {          // Result<KZGSetup, SetupFileError>
  bool ok;
  std::unique_ptr<KZGSetup> setup;
  ptau::SetupFileError error;
};
FileSetup new_from_file(std::shared_ptr<Device> dev, const std::string& path);
}  // namespace kzg
}  // namespace keaki
