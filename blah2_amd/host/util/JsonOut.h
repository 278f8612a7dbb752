// Minimal JSON emitter for the Map / Detection products.
//
// The reference serialises with rapidjson 1.1.0 (lib/vcpkg.json:6,15), a
// Writer<StringBuffer> with SetMaxDecimalPlaces(2) (src/data/Map.cpp:157-160,
// src/data/Detection.cpp:79-82).  rapidjson is not available here, so the
// relevant part of its published number formatting is restated:
//   * integers print as plain decimal,
//   * doubles go through dtoa(value, maxDecimalPlaces): shortest round-trip
//     digits, then "Prettify", which TRUNCATES (never rounds) after
//     maxDecimalPlaces decimals, strips trailing zeros but keeps one, prints
//     |v| < 10^-maxDecimalPlaces as 0.0 and keeps at least ".0",
//   * NaN/Inf make Writer::Double fail (the document is cut short there);
//     here they are written as null and flagged in ok().
// Digits come from std::to_chars (shortest round-trip); rapidjson's Grisu2
// yields the same digits except in the rare cases where Grisu2 is not
// shortest, which the 2-decimal truncation almost always hides.
#ifndef BLAH2HIP_HOST_JSONOUT_H
#define BLAH2HIP_HOST_JSONOUT_H

#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>

namespace blah2json {

inline void append_exponent(std::string &out, int K)
{
  if (K < 0) { out.push_back('-'); K = -K; }
  out += std::to_string(K);
}

// digits[0..len) * 10^k  ->  rapidjson Prettify
inline void prettify(std::string &out, const char *digits, int len, int k, int maxDec)
{
  const int kk = len + k; // 10^(kk-1) <= v < 10^kk
  if (0 <= k && kk <= 21) {
    out.append(digits, len);
    out.append((size_t)(kk - len), '0');
    out += ".0";
  } else if (0 < kk && kk <= 21) {
    std::string s(digits, kk);
    s.push_back('.');
    s.append(digits + kk, len - kk);
    if (0 > k + maxDec) {
      // truncate to maxDec decimals, drop trailing zeros but keep one decimal
      int end = kk + maxDec; // index of last kept decimal in s
      while (end > kk + 1 && s[end] == '0') end--;
      s.resize(end + 1);
    }
    out += s;
  } else if (-6 < kk && kk <= 0) {
    std::string s = "0.";
    s.append((size_t)(-kk), '0');
    s.append(digits, len);
    if (len - kk > maxDec) {
      int end = maxDec + 1;
      while (end > 2 && s[end] == '0') end--;
      s.resize(end + 1);
    }
    out += s;
  } else if (kk < -maxDec) {
    out += "0.0";
  } else if (len == 1) {
    out.push_back(digits[0]);
    out.push_back('e');
    append_exponent(out, kk - 1);
  } else {
    out.push_back(digits[0]);
    out.push_back('.');
    out.append(digits + 1, len - 1);
    out.push_back('e');
    append_exponent(out, kk - 1);
  }
}

// returns false for NaN/Inf (writes null)
inline bool write_double(std::string &out, double v, int maxDec = 2)
{
  if (std::isnan(v) || std::isinf(v)) { out += "null"; return false; }
  if (v == 0.0) { out += std::signbit(v) ? "-0.0" : "0.0"; return true; }
  if (v < 0) { out.push_back('-'); v = -v; }
  char buf[64];
  auto r = std::to_chars(buf, buf + sizeof buf - 1, v, std::chars_format::scientific);
  *r.ptr = '\0'; // to_chars does not terminate; the exponent is parsed with atoi below
  // d[.ddd]e[+-]XX  ->  digit string + decimal exponent
  char digits[32];
  int len = 0, exp10 = 0;
  const char *p = buf;
  for (; p < r.ptr && *p != 'e'; p++)
    if (*p != '.') digits[len++] = *p;
  if (p < r.ptr) exp10 = std::atoi(p + 1);
  while (len > 1 && digits[len - 1] == '0') len--;
  prettify(out, digits, len, exp10 - (len - 1), maxDec);
  return true;
}

class Writer {
public:
  explicit Writer(int maxDecimalPlaces = 2) : dec_(maxDecimalPlaces) {}
  void begin_object() { sep(); s_.push_back('{'); first_ = true; }
  void end_object() { s_.push_back('}'); first_ = false; }
  void begin_array() { sep(); s_.push_back('['); first_ = true; }
  void end_array() { s_.push_back(']'); first_ = false; }
  void key(const char *k) { sep(); s_.push_back('"'); s_ += k; s_ += "\":"; first_ = true; }
  void value(double v) { sep(); if (!write_double(s_, v, dec_)) ok_ = false; }
  void value(int64_t v) { sep(); s_ += std::to_string(v); }
  void value(uint64_t v) { sep(); s_ += std::to_string(v); }
  void value(int v) { value((int64_t)v); }
  void value(uint32_t v) { value((uint64_t)v); }
  const std::string &str() const { return s_; }
  bool ok() const { return ok_; }

private:
  void sep() { if (!first_) s_.push_back(','); first_ = false; }
  std::string s_;
  int dec_;
  bool first_ = true;
  bool ok_ = true;
};

// Replaces the array value of `"key":[ ... ]` in a flat JSON object string.
inline std::string replace_array(const std::string &json, const char *key, const std::string &arr)
{
  const std::string pat = std::string("\"") + key + "\":[";
  const size_t a = json.find(pat);
  if (a == std::string::npos) return json;
  const size_t b = json.find(']', a);
  if (b == std::string::npos) return json;
  return json.substr(0, a + pat.size() - 1) + arr + json.substr(b + 1);
}

} // namespace blah2json
#endif
