// host_settings.cpp -- readers and writers for the files OCTproZ itself writes, so that a reference
// installation's settings and calibration curves drive this pipeline unchanged:
//   * settings.ini (QSettings::IniFormat through SettingsFileManager, src/settingsfilemanager.cpp:28-112): groups [processing],
//     [streaming], [record] and the acquisition plug-in's group ("Virtual%20OCT%20System"); key names = the PROC_* / STREAM_* /
//     REC_* macros of octproz_project/octproz/src/sidebar.h:47-94; value conversions as Sidebar::loadSettings
//     (src/sidebar.cpp:173-229: toBool / toUInt / toDouble / toString) and VirtualOCTSystemSettingsDialog::setSettings
//     (virtualoctsystemsettingsdialog.cpp:43-56: toInt / toBool / toString); spin-box doubles are assigned to float fields.
//   * curve CSV ("resampling.csv", "background.csv"): one header line, then "index;value" per line, value = field 1 of a ';'
//     split parsed with QString::toFloat (src/octalgorithmparametersmanager.cpp:12-30), written with QTextStream << float (:32-45).
// No Qt here.  The INI syntax, the string -> bool / integer / double rules and the number formats below restate what Qt 5 does;
// they are pinned by vectors captured from the reference's own classes running on this image's Qt 5.9.7
// (tests/golden/host_ref.json, tests/test_host_reference.py): hand-written files both sides must read alike, files the reference
// wrote, and files written here that the reference read back.
#include <cctype>
#include <cerrno>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/octhost.h"

namespace {

// ---------------------------------------------------------------- text helpers
void appendUtf8(std::string& out, unsigned cp) {
	if (cp < 0x80) out.push_back((char)cp);
	else if (cp < 0x800) { out.push_back((char)(0xC0 | (cp >> 6))); out.push_back((char)(0x80 | (cp & 0x3F))); }
	else if (cp < 0x10000) { out.push_back((char)(0xE0 | (cp >> 12))); out.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); out.push_back((char)(0x80 | (cp & 0x3F))); }
	else { out.push_back((char)(0xF0 | ((cp >> 18) & 7))); out.push_back((char)(0x80 | ((cp >> 12) & 0x3F))); out.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); out.push_back((char)(0x80 | (cp & 0x3F))); }
}
// UTF-8 -> code points (invalid bytes are taken as Latin-1)
std::vector<unsigned> decodeUtf8(const std::string& s) {
	std::vector<unsigned> cps;
	for (size_t i = 0; i < s.size();) {
		const unsigned char c = (unsigned char)s[i];
		int n = c < 0x80 ? 0 : (c >> 5) == 6 ? 1 : (c >> 4) == 14 ? 2 : (c >> 3) == 30 ? 3 : -1;
		unsigned cp = n <= 0 ? c : (c & (0x3F >> n));
		bool ok = n >= 0 && i + (size_t)n < s.size() + 0;
		for (int k = 1; ok && k <= n; ++k) { const unsigned char d = (unsigned char)s[i + k]; if ((d >> 6) != 2) ok = false; else cp = (cp << 6) | (d & 0x3F); }
		if (!ok) { cps.push_back(c); ++i; } else { cps.push_back(cp); i += (size_t)n + 1; }
	}
	return cps;
}
bool isSpace(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\v' || c == '\f'; }
std::string trimmed(const std::string& s) {
	size_t a = 0, b = s.size();
	while (a < b && isSpace(s[a])) ++a;
	while (b > a && isSpace(s[b - 1])) --b;
	return s.substr(a, b - a);
}
int hexVal(char c) { return c >= '0' && c <= '9' ? c - '0' : c >= 'a' && c <= 'f' ? c - 'a' + 10 : c >= 'A' && c <= 'F' ? c - 'A' + 10 : -1; }

// ---------------------------------------------------------------- QString -> number (C locale)
// QString::toDouble: optional white space around ONE number in decimal notation ("+3", ".5", "1e3"), or inf / nan; no hexadecimal
// form, no decimal comma, no trailing text -- anything else is 0 (ok = false)
bool qtToDouble(const std::string& text, double* out) {
	const std::string s = trimmed(text);
	*out = 0.0;
	if (s.empty()) return false;
	size_t i = 0;
	if (s[i] == '+' || s[i] == '-') ++i;
	std::string rest;
	for (size_t k = i; k < s.size(); ++k) rest.push_back((char)tolower((unsigned char)s[k]));
	if (rest == "inf") { *out = s[0] == '-' ? -HUGE_VAL : HUGE_VAL; return true; }
	if (rest == "nan") {  // (the quiet NaN Qt 5.9 hands out: as a float 0x7FE00000, tests/golden/host_ref.json)
		const unsigned long long bits = 0x7FFC000000000000ull;
		memcpy(out, &bits, sizeof bits);
		return true;
	}
	size_t digits = 0;
	while (i < s.size() && isdigit((unsigned char)s[i])) { ++i; ++digits; }
	if (i < s.size() && s[i] == '.') { ++i; while (i < s.size() && isdigit((unsigned char)s[i])) { ++i; ++digits; } }
	if (digits == 0) return false;
	if (i < s.size() && (s[i] == 'e' || s[i] == 'E')) {
		++i;
		if (i < s.size() && (s[i] == '+' || s[i] == '-')) ++i;
		size_t ed = 0;
		while (i < s.size() && isdigit((unsigned char)s[i])) { ++i; ++ed; }
		if (ed == 0) return false;
	}
	if (i != s.size()) return false;
	*out = strtod(s.c_str(), nullptr);
	return true;
}
// QString::toFloat = toDouble, then QLocaleData::convertDoubleToFloat: a finite value beyond the float range, or one that
// underflows to zero, gives 0 (ok = false) -- not infinity
float qtToFloat(const std::string& text) {
	double d;
	if (!qtToDouble(text, &d)) return 0.0f;
	if (std::isinf(d) || std::isnan(d)) return (float)d;
	if (std::fabs(d) > (double)FLT_MAX) return 0.0f;
	const float f = (float)d;
	if (d != 0.0 && f == 0.0f) return 0.0f;
	return f;
}
// QVariant(QString)::toInt / toUInt (qvariant.cpp qConvertToNumber / qConvertToUnsignedNumber): QString::toLongLong resp.
// toULongLong -- white space, sign, decimal digits, nothing else ("33.0", "1e1", "0x10" fail: 0; toULongLong refuses a minus sign) --
// and the 64-bit result is then CUT to 32 bits without a range check ("4000000000" as int is -294967296)
bool qtToInteger(const std::string& text, bool isUnsigned, long long* out) {
	const std::string s = trimmed(text);
	*out = 0;
	size_t i = 0;
	if (i < s.size() && (s[i] == '+' || s[i] == '-')) { if (isUnsigned && s[i] == '-') return false; ++i; }
	if (i == s.size()) return false;
	for (size_t k = i; k < s.size(); ++k) if (!isdigit((unsigned char)s[k])) return false;
	errno = 0;
	if (isUnsigned) {
		const unsigned long long v = strtoull(s.c_str(), nullptr, 10);
		if (errno) return false;
		*out = (long long)(uint32_t)v;
	} else {
		const long long v = strtoll(s.c_str(), nullptr, 10);
		if (errno) return false;
		*out = (long long)(int32_t)(uint32_t)(unsigned long long)v;
	}
	return true;
}

// ---------------------------------------------------------------- numbers -> text
// %g with the exponent written as Qt writes it (at least one digit: "1e-7", "1e+21"); -0 keeps its sign
std::string tidyExponent(std::string s) {
	const size_t e = s.find('e');
	if (e == std::string::npos || e + 2 >= s.size()) return s;
	size_t d = e + 2;
	while (d + 1 < s.size() && s[d] == '0') s.erase(d, 1);
	return s;
}
// QTextStream << float (realNumberNotation SmartNotation, precision 6): the "%g" form -- 1e-07, 123457, 1.67772e+07, nan, inf,
// -inf -- with two differences from printf: minus zero comes out as "0", and a value exactly half way between two 6-digit
// numbers is rounded UP in magnitude (1020.125 -> "1020.13"; printf rounds the tie to even, "1020.12")
std::string qtStreamFloat(float v) {
	if (std::isnan(v)) return "nan";
	if (std::isinf(v)) return v < 0 ? "-inf" : "inf";
	if (v == 0.0f) return "0";
	// the exact decimal expansion of the binary value (a float has at most 112 significant decimal digits)
	char e[200];
	snprintf(e, sizeof e, "%.120e", std::fabs((double)v));  // d.ddddd...e[+-]xx
	std::string digits;
	int exp10 = 0;
	{
		const char* p = e;
		digits.push_back(*p++);
		if (*p == '.') ++p;
		while (*p && *p != 'e') digits.push_back(*p++);
		exp10 = atoi(p + 1);
	}
	const int P = 6;
	bool up = digits[P] >= '5';  // ties go up: the digits behind position P are exact, so '5' followed by zeros is the tie
	std::string d6 = digits.substr(0, P);
	if (up) {
		int k = P - 1;
		while (k >= 0 && d6[(size_t)k] == '9') d6[(size_t)k--] = '0';
		if (k >= 0) d6[(size_t)k]++;
		else { d6 = "1" + d6.substr(0, P - 1); ++exp10; }
	}
	while (d6.size() > 1 && d6.back() == '0') d6.pop_back();  // %g drops trailing zeros
	std::string r = v < 0 ? "-" : "";
	if (exp10 < -4 || exp10 >= P) {
		r += d6.substr(0, 1);
		if (d6.size() > 1) r += "." + d6.substr(1);
		char x[16];
		snprintf(x, sizeof x, "e%c%02d", exp10 < 0 ? '-' : '+', std::abs(exp10));
		r += x;
	} else if (exp10 < 0) {
		r += "0." + std::string((size_t)(-exp10 - 1), '0') + d6;
	} else {
		if ((int)d6.size() <= exp10 + 1) r += d6 + std::string((size_t)(exp10 + 1 - (int)d6.size()), '0');
		else r += d6.substr(0, (size_t)exp10 + 1) + "." + d6.substr((size_t)exp10 + 1);
	}
	return r;
}
// QVariant(double) -> QString as QSettings stores it: the shortest text that reads back to the same value (QString::number(d, 'g',
// FloatingPointShortest)); the fields here are float32, so: the shortest text that reads back to the same FLOAT, which is what a
// spin box shows for it ("0.535239", "0.95", "1e-7")
std::string shortestFloat(float v) {
	if (std::isnan(v)) return "nan";
	if (std::isinf(v)) return v < 0 ? "-inf" : "inf";
	char b[64];
	for (int prec = 1; prec <= 9; ++prec) {
		snprintf(b, sizeof b, "%.*g", prec, (double)v);
		if (strtof(b, nullptr) == v) break;
	}
	return tidyExponent(b);
}

// ---------------------------------------------------------------- QSettings INI syntax
// iniEscapedString (qsettings.cpp): backslash escapes for control characters, '"' and '\\'; everything outside ASCII as \x<hex> (the
// default, codec-less form: what OCTproZ's QSettings writes); a hexadecimal digit behind a \x escape is escaped too; a value that
// contains ';' ',' '=' or starts / ends with a blank is put in double quotes; a leading '@' is doubled (variantToString)
std::string iniEscapedString(const std::string& utf8) {
	std::vector<unsigned> cps = decodeUtf8(utf8);
	if (!cps.empty() && cps[0] == '@') cps.insert(cps.begin(), '@');
	std::string r;
	bool needsQuotes = false, escapeNextIfDigit = false;
	for (unsigned ch : cps) {
		if (ch == ';' || ch == ',' || ch == '=') needsQuotes = true;
		char hex[16];
		if (escapeNextIfDigit && ch < 128 && hexVal((char)ch) >= 0) { snprintf(hex, sizeof hex, "\\x%x", ch); r += hex; continue; }
		escapeNextIfDigit = false;
		switch (ch) {
		case 0: r += "\\0"; escapeNextIfDigit = true; break;
		case '\a': r += "\\a"; break;
		case '\b': r += "\\b"; break;
		case '\f': r += "\\f"; break;
		case '\n': r += "\\n"; break;
		case '\r': r += "\\r"; break;
		case '\t': r += "\\t"; break;
		case '\v': r += "\\v"; break;
		case '"': case '\\': r.push_back('\\'); r.push_back((char)ch); break;
		default:
			if (ch <= 0x1F || ch >= 0x7F) { snprintf(hex, sizeof hex, "\\x%x", ch); r += hex; escapeNextIfDigit = true; }
			else r.push_back((char)ch);
		}
	}
	if (needsQuotes || (!r.empty() && (r.front() == ' ' || r.back() == ' '))) r = "\"" + r + "\"";
	return r;
}
// iniEscapedKey: letters, digits, '_' '-' '.' stay, '/' separates groups, everything else %XX (%uXXXX beyond Latin-1)
std::string iniEscapedKey(const std::string& utf8) {
	std::string r;
	char b[16];
	for (unsigned ch : decodeUtf8(utf8)) {
		if (ch == '/' || (ch < 128 && (isalnum((int)ch) || ch == '_' || ch == '-' || ch == '.'))) r.push_back((char)ch);
		else if (ch <= 0xFF) { snprintf(b, sizeof b, "%%%02X", ch); r += b; }
		else { snprintf(b, sizeof b, "%%U%04X", ch); r += b; }
	}
	return r;
}
std::string iniUnescapedKey(const std::string& s) {
	std::string out;
	for (size_t i = 0; i < s.size(); ++i) {
		if (s[i] == '%' && i + 2 < s.size() + 0 && (s[i + 1] == 'u' || s[i + 1] == 'U') && i + 5 < s.size() + 0) {
			unsigned cp = 0; bool ok = true;
			for (int k = 2; k <= 5; ++k) { const int h = hexVal(s[i + (size_t)k]); if (h < 0) ok = false; else cp = cp * 16 + (unsigned)h; }
			if (ok) { appendUtf8(out, cp); i += 5; continue; }
		}
		if (s[i] == '%' && i + 2 < s.size() + 0 && hexVal(s[i + 1]) >= 0 && hexVal(s[i + 2]) >= 0) {
			appendUtf8(out, (unsigned)(hexVal(s[i + 1]) * 16 + hexVal(s[i + 2])));
			i += 2;
			continue;
		}
		appendUtf8(out, (unsigned char)s[i]);  // (file bytes are Latin-1)
	}
	return out;
}

struct Value { std::string text; bool isList = false; };  // a list (unquoted commas) joined with ','

// iniUnescapedStringList (qsettings.cpp): the value part of a line -> string or string list.  Leading blanks skipped, trailing
// blanks of an unquoted value chopped, "..." protects ; , = and blanks, backslash escapes (\a \b \f \n \r \t \v \" \? \' \\,
// \x<hex digits>, \<octal digits>, backslash + line end = continuation, any other escaped character is dropped), an unquoted ','
// separates list items.  File bytes are Latin-1; the result is UTF-8.
Value iniUnescapedStringList(const std::string& str) {
	Value res;
	std::string cur;                 // current item (UTF-8)
	std::vector<std::string> items;
	bool inQuoted = false, curQuoted = false;
	size_t i = 0, chopLimit = 0;
	const size_t to = str.size();
	auto chop = [&](std::string& s, size_t limit) { while (s.size() > limit && (s.back() == ' ' || s.back() == '\t')) s.pop_back(); };
	auto skipSpaces = [&]() { while (i < to && (str[i] == ' ' || str[i] == '\t')) ++i; };
	skipSpaces();
	chopLimit = cur.size();
	while (i < to) {
		const char c = str[i];
		if (c == '\\') {
			++i;
			if (i >= to) break;
			const char ch = str[i++];
			const char* codes = "abfnrtv\"?'\\";
			const char* mapped = "\a\b\f\n\r\t\v\"?'\\";
			const char* hit = strchr(codes, ch);
			if (ch && hit) { cur.push_back(mapped[hit - codes]); }
			else if (ch == 'x') {
				unsigned v = 0; bool any = false;
				while (i < to && hexVal(str[i]) >= 0) { v = (v << 4) + (unsigned)hexVal(str[i]); ++i; any = true; }
				if (any) appendUtf8(cur, v & 0xFFFFu);
			} else if (ch >= '0' && ch <= '7') {
				unsigned v = (unsigned)(ch - '0');
				while (i < to && str[i] >= '0' && str[i] <= '7') { v = (v << 3) + (unsigned)(str[i] - '0'); ++i; }
				appendUtf8(cur, v & 0xFFFFu);
			} else if (ch == '\n' || ch == '\r') {
				if (i < to && (str[i] == '\n' || str[i] == '\r') && str[i] != ch) ++i;
			}  // any other character behind a backslash is skipped
			chopLimit = cur.size();
		} else if (c == '"') {
			++i;
			curQuoted = true;
			inQuoted = !inQuoted;
			if (!inQuoted) skipSpaces();
		} else if (c == ',' && !inQuoted) {
			if (!curQuoted) chop(cur, chopLimit);
			res.isList = true;
			items.push_back(cur);
			cur.clear();
			curQuoted = false;
			++i;
			skipSpaces();
			chopLimit = 0;
		} else {
			size_t j = i + 1;
			while (j < to && str[j] != '\\' && str[j] != '"' && str[j] != ',') ++j;
			for (size_t k = i; k < j; ++k) appendUtf8(cur, (unsigned char)str[k]);
			i = j;
		}
	}
	if (!curQuoted) chop(cur, chopLimit);
	if (res.isList) {
		items.push_back(cur);
		for (size_t k = 0; k < items.size(); ++k) { if (k) res.text.push_back(','); res.text += items[k]; }
	} else {
		res.text = cur;
	}
	// variant prefixes (stringToVariant): "@@..." is a string that starts with '@'; @Variant(<QDataStream Qt_4_0 bytes>) carries a
	// typed value -- the numeric ones are decoded (a float in a settings map is stored this way: type 135, four bytes, big endian);
	// @ByteArray(...), @Invalid(), the geometry forms and anything undecodable are no strings: numbers and booleans read from them
	// are 0 / false, as from an empty string
	if (!res.isList && res.text.size() >= 2 && res.text[0] == '@') {
		if (res.text[1] == '@') res.text.erase(0, 1);
		else if (res.text.back() == ')') {
			std::string decoded;
			if (res.text.compare(0, 9, "@Variant(") == 0) {
				std::vector<unsigned> cps = decodeUtf8(res.text.substr(9, res.text.size() - 10));
				std::vector<unsigned char> b;
				for (unsigned cp : cps) b.push_back((unsigned char)(cp & 0xFF));
				auto be = [&](size_t at, int n) { unsigned long long v = 0; for (int k = 0; k < n; ++k) v = (v << 8) | b[at + (size_t)k]; return v; };
				if (b.size() >= 4) {
					const unsigned long long type = be(0, 4);
					char t[64] = "";
					if (type == 135 && b.size() == 8) { const uint32_t u = (uint32_t)be(4, 4); float f; memcpy(&f, &u, 4); snprintf(t, sizeof t, "%.9g", (double)f); }
					else if (type == 6 && b.size() == 12) { const unsigned long long u = be(4, 8); double d; memcpy(&d, &u, 8); snprintf(t, sizeof t, "%.17g", d); }
					else if (type == 2 && b.size() == 8) snprintf(t, sizeof t, "%d", (int)(int32_t)(uint32_t)be(4, 4));
					else if (type == 3 && b.size() == 8) snprintf(t, sizeof t, "%u", (unsigned)be(4, 4));
					else if (type == 1 && b.size() == 5) snprintf(t, sizeof t, "%s", b[4] ? "true" : "false");
					decoded = t;
				}
			}
			res.text = decoded;
		}
	}
	return res;
}

typedef std::map<std::string, std::map<std::string, Value>> Ini;

// readIniLine / readIniFile (qsettings.cpp): a line ends at a line break outside quotes, a backslash takes the next character (and
// a line break) with it, ';' outside quotes starts a comment (at the start of a line: the whole line), the first '=' outside quotes
// separates key and value; "[General]" (any case) is the group-less section; section and key names are case sensitive on this
// platform, later assignments win
bool parseIni(const char* path, Ini& ini) {
	std::ifstream f(path, std::ios::binary);
	if (!f) return false;
	std::stringstream ss;
	ss << f.rdbuf();
	const std::string data = ss.str();
	const size_t n = data.size();
	size_t pos = 0;
	if (n >= 3 && (unsigned char)data[0] == 0xEF && (unsigned char)data[1] == 0xBB && (unsigned char)data[2] == 0xBF) pos = 3;
	std::string section;  // "" = [General] / before any section; otherwise with a trailing '/'
	while (pos < n) {
		size_t lineStart = pos, i = pos;
		long equalsPos = -1;
		bool inQuotes = false;
		size_t lineEnd = n;
		bool done = false;
		while (i < n && !done) {
			const char ch = data[i++];
			switch (ch) {
			case '"': inQuotes = !inQuotes; break;
			case '=': if (!inQuotes && equalsPos < 0) equalsPos = (long)(i - 1); break;
			case '\n': case '\r':
				if (i == lineStart + 1) ++lineStart;
				else if (!inQuotes) { lineEnd = i - 1; done = true; }
				break;
			case '\\':
				if (i < n) {
					const char c1 = data[i++];
					if (i < n) { const char c2 = data[i]; if ((c1 == '\n' && c2 == '\r') || (c1 == '\r' && c2 == '\n')) ++i; }
				}
				break;
			case ';':
				if (i == lineStart + 1) {  // a comment line
					while (i < n && data[i] != '\n' && data[i] != '\r') ++i;
					lineStart = i;
				} else if (!inQuotes) {
					lineEnd = i - 1;
					done = true;
					while (i < n && data[i] != '\n' && data[i] != '\r') ++i;  // the comment itself
				}
				break;
			default: break;
			}
		}
		if (!done) lineEnd = i;
		pos = i;
		if (lineEnd <= lineStart) continue;
		const std::string line = data.substr(lineStart, lineEnd - lineStart);
		const std::string lt = trimmed(line);
		if (lt.empty()) continue;
		if (lt[0] == '[') {
			const size_t close = lt.find(']');
			std::string name = trimmed(lt.substr(1, close == std::string::npos ? std::string::npos : close - 1));
			std::string lower;
			for (char c : name) lower.push_back((char)tolower((unsigned char)c));
			if (lower == "general") section.clear();
			else if (lower == "%general") section = name.substr(1) + "/";
			else section = iniUnescapedKey(name) + "/";
			continue;
		}
		if (equalsPos < 0 || (size_t)equalsPos < lineStart || (size_t)equalsPos >= lineEnd) continue;
		const std::string key = iniUnescapedKey(trimmed(data.substr(lineStart, (size_t)equalsPos - lineStart)));
		const Value v = iniUnescapedStringList(data.substr((size_t)equalsPos + 1, lineEnd - (size_t)equalsPos - 1));
		std::string grp = section.empty() ? std::string() : section.substr(0, section.size() - 1);
		ini[grp][key] = v;
	}
	return true;
}

// QVariant conversions of a value read from the file (a QString, or a QStringList where the text had unquoted commas)
struct Group {
	const std::map<std::string, Value>* m;
	bool has(const char* k) const { return m && m->count(k); }
	// QVariant(QString)::toBool: everything but "", "0" and "false" (any case) is true; a list is false
	bool b(const char* k, bool d) const {
		if (!has(k)) return d;
		const Value& v = m->at(k);
		if (v.isList) return false;
		std::string s;
		for (char c : v.text) s.push_back((char)tolower((unsigned char)c));
		return !(s.empty() || s == "0" || s == "false");
	}
	double num(const char* k, double d) const {  // toDouble
		if (!has(k)) return d;
		const Value& v = m->at(k);
		double x = 0.0;
		if (!v.isList) qtToDouble(v.text, &x);
		return x;
	}
	long long integer(const char* k, long long d, bool isUnsigned) const {  // toUInt / toInt: 0 unless the text is an integer
		if (!has(k)) return d;
		const Value& v = m->at(k);
		long long x = 0;
		if (!v.isList) qtToInteger(v.text, isUnsigned, &x);
		return x;
	}
	long long u32(const char* k, long long d) const { return integer(k, d, true); }
	long long i32(const char* k, long long d) const { return integer(k, d, false); }
	std::string str(const char* k) const { return has(k) ? m->at(k).text : std::string(); }
};

Group group(const Ini& ini, const char* name) {
	auto it = ini.find(name);
	return Group{it == ini.end() ? nullptr : &it->second};
}

void copyPath(char* dst, size_t n, const std::string& s) {
	if (!dst || n == 0) return;
	std::snprintf(dst, n, "%s", s.c_str());
}

}  // namespace

extern "C" {

int octhost_load_settings_ini(const char* path, OctPipeParams* params, OctHostCurveSettings* curves,
                              OctHostVirtualParams* vsys, char* vsysFilePath, size_t vsysFilePathSize) {
	if (!path || !params) return OCTPIPE_ERR_INVALID_ARGUMENT;
	Ini ini;
	if (!parseIni(path, ini)) return OCTPIPE_ERR_INVALID_ARGUMENT;
	const Group p = group(ini, "processing"), s = group(ini, "streaming"), r = group(ini, "record");
	// Sidebar::loadSettings (sidebar.cpp:186-229) -> Sidebar::updateProcessingParams (sidebar.cpp:319-338)
	params->bitshift = p.b("bitshift", params->bitshift);
	params->bscanFlip = p.b("flip_bscans", params->bscanFlip);
	params->signalLogScaling = p.b("log", params->signalLogScaling);
	params->signalGrayscaleMax = (float)p.num("max", params->signalGrayscaleMax);
	params->signalGrayscaleMin = (float)p.num("min", params->signalGrayscaleMin);
	params->signalMultiplicator = (float)p.num("coeff", params->signalMultiplicator);
	params->signalAddend = (float)p.num("addend", params->signalAddend);
	params->fixedPatternNoiseRemoval = p.b("fixed_pattern_removal", params->fixedPatternNoiseRemoval);
	params->continuousFixedPatternNoiseDetermination = p.b("fixed_pattern_removal_continuously", params->continuousFixedPatternNoiseDetermination);
	params->bscansForNoiseDetermination = (uint32_t)p.u32("fixed_pattern_removal_bscans", params->bscansForNoiseDetermination);
	params->sinusoidalScanCorrection = p.b("sinusoidal_scan_correction", params->sinusoidalScanCorrection);
	params->rollingAverageWindowSize = (int32_t)p.u32("background_removal_window_size", params->rollingAverageWindowSize);
	params->backgroundRemoval = p.b("background_removal", params->backgroundRemoval);
	params->postProcessBackgroundRemoval = p.b("post_processing_background_removal", params->postProcessBackgroundRemoval);
	params->postProcessBackgroundWeight = (float)p.num("post_processing_background_removal_weight", params->postProcessBackgroundWeight);
	params->postProcessBackgroundOffset = (float)p.num("post_processing_background_removal_offset", params->postProcessBackgroundOffset);
	// updateResamplingParams / updateDispersionParams / updateWindowingParams, sidebar.cpp:372-430
	params->resampling = p.b("resampling", params->resampling);
	params->resamplingInterpolation = (int32_t)p.u32("resampling_interpolation", params->resamplingInterpolation);
	params->dispersionCompensation = p.b("dispersion_compensation", params->dispersionCompensation);
	params->windowing = p.b("windowing", params->windowing);
	// updateStreamingParams :340-345, updateRecordingParams :347-360
	params->streamToHost = s.b("streaming_enabled", params->streamToHost);
	params->streamingBuffersToSkip = (uint32_t)s.u32("streaming_skip", params->streamingBuffersToSkip);
	params->streamFloatToHost = r.b("save_as_32_bit_float", params->streamFloatToHost);
	if (curves) {
		curves->c[0] = (float)p.num("resampling_c0", curves->c[0]); curves->c[1] = (float)p.num("resampling_c1", curves->c[1]);
		curves->c[2] = (float)p.num("resampling_c2", curves->c[2]); curves->c[3] = (float)p.num("resampling_c3", curves->c[3]);
		curves->d[0] = (float)p.num("dispersion_compensation_d0", curves->d[0]); curves->d[1] = (float)p.num("dispersion_compensation_d1", curves->d[1]);
		curves->d[2] = (float)p.num("dispersion_compensation_d2", curves->d[2]); curves->d[3] = (float)p.num("dispersion_compensation_d3", curves->d[3]);
		curves->windowType = (int32_t)p.u32("window_type", curves->windowType);
		curves->windowCenter = (float)p.num("window_center_position", curves->windowCenter);
		curves->windowFillFactor = (float)p.num("window_fill_factor", curves->windowFillFactor);
		curves->customResampling = p.b("custom_resampling", curves->customResampling);
		copyPath(curves->customResamplingFilePath, sizeof(curves->customResamplingFilePath), p.str("custom_resampling_filepath"));
		copyPath(curves->postBackgroundFilePath, sizeof(curves->postBackgroundFilePath), p.str("post_processing_background_filepath"));
	}
	if (vsys) {  // VirtualOCTSystemSettingsDialog::setSettings (virtualoctsystemsettingsdialog.cpp:43-56), group = plug-in name
		const Group v = group(ini, "Virtual OCT System");
		vsys->bitDepth = (unsigned)v.i32("bit_depth", vsys->bitDepth);
		vsys->width = (unsigned)v.i32("width", vsys->width);
		vsys->height = (unsigned)v.i32("height", vsys->height);
		vsys->depth = (unsigned)v.i32("depth", vsys->depth);
		vsys->buffersPerVolume = (unsigned)v.i32("buffers_per_volume", vsys->buffersPerVolume);
		vsys->buffersFromFile = (unsigned)v.i32("buffers_from_file", vsys->buffersFromFile);
		vsys->bscanOffset = (unsigned)v.i32("bscan_offset", vsys->bscanOffset);
		vsys->waitTimeUs = (unsigned)v.i32("wait_time", vsys->waitTimeUs);
		vsys->copyFileToRam = v.b("copy_file_to_ram", vsys->copyFileToRam != 0);
		vsys->syncWithProcessing = v.b("sync_with_processing", vsys->syncWithProcessing != 0);
		copyPath(vsysFilePath, vsysFilePathSize, v.str("file_path"));
	}
	return OCTPIPE_OK;
}

// Writer of the same file (the Recorder's "save meta info" leg stores the settings next to a recording, and OCTproZ reads the
// file back at start-up): the keys octhost_load_settings_ini understands, in QSettings' own syntax -- true / false, the
// shortest number text that reads back to the same value, strings escaped and quoted by iniEscapedString (paths with
// backslashes, commas, semicolons, non-ASCII characters), group names by iniEscapedKey.
int octhost_save_settings_ini(const char* path, const OctPipeParams* params, const OctHostCurveSettings* curves,
                              const OctHostVirtualParams* vsys, const char* vsysFilePath, const char* timestamp) {
	if (!path || !params) return OCTPIPE_ERR_INVALID_ARGUMENT;
	FILE* f = fopen(path, "wb");
	if (!f) return OCTPIPE_ERR_INVALID_ARGUMENT;
	auto B = [](int v) { return v ? "true" : "false"; };
	auto F = [](float v) { return shortestFloat(v); };
	auto S = [](const char* s) { return iniEscapedString(s ? s : ""); };
	fprintf(f, "[General]\ntimestamp=%s\n\n", S(timestamp).c_str());
	fprintf(f, "[record]\nsave_as_32_bit_float=%s\n\n", B(params->streamFloatToHost));
	fprintf(f, "[processing]\n");
	fprintf(f, "addend=%s\nbitshift=%s\ncoeff=%s\n", F(params->signalAddend).c_str(), B(params->bitshift), F(params->signalMultiplicator).c_str());
	fprintf(f, "dispersion_compensation=%s\n", B(params->dispersionCompensation));
	if (curves) for (int i = 0; i < 4; ++i) fprintf(f, "dispersion_compensation_d%d=%s\n", i, F(curves->d[i]).c_str());
	fprintf(f, "fixed_pattern_removal=%s\nfixed_pattern_removal_continuously=%s\nfixed_pattern_removal_bscans=%u\n", B(params->fixedPatternNoiseRemoval),
	        B(params->continuousFixedPatternNoiseDetermination), params->bscansForNoiseDetermination);
	fprintf(f, "flip_bscans=%s\nlog=%s\nmax=%s\nmin=%s\n", B(params->bscanFlip), B(params->signalLogScaling), F(params->signalGrayscaleMax).c_str(), F(params->signalGrayscaleMin).c_str());
	fprintf(f, "resampling=%s\n", B(params->resampling));
	if (curves) for (int i = 0; i < 4; ++i) fprintf(f, "resampling_c%d=%s\n", i, F(curves->c[i]).c_str());
	fprintf(f, "resampling_interpolation=%d\nsinusoidal_scan_correction=%s\n", params->resamplingInterpolation, B(params->sinusoidalScanCorrection));
	if (curves) fprintf(f, "window_center_position=%s\nwindow_fill_factor=%s\nwindow_type=%d\n", F(curves->windowCenter).c_str(), F(curves->windowFillFactor).c_str(), curves->windowType);
	fprintf(f, "windowing=%s\nbackground_removal=%s\nbackground_removal_window_size=%d\n", B(params->windowing), B(params->backgroundRemoval), params->rollingAverageWindowSize);
	if (curves) fprintf(f, "custom_resampling=%s\ncustom_resampling_filepath=%s\npost_processing_background_filepath=%s\n", B(curves->customResampling),
	                    S(curves->customResamplingFilePath).c_str(), S(curves->postBackgroundFilePath).c_str());
	fprintf(f, "post_processing_background_removal=%s\npost_processing_background_removal_offset=%s\npost_processing_background_removal_weight=%s\n\n",
	        B(params->postProcessBackgroundRemoval), F(params->postProcessBackgroundOffset).c_str(), F(params->postProcessBackgroundWeight).c_str());
	fprintf(f, "[streaming]\nstreaming_enabled=%s\nstreaming_skip=%u\n\n", B(params->streamToHost), params->streamingBuffersToSkip);
	if (vsys) {
		fprintf(f, "[%s]\nbit_depth=%u\nbuffers_from_file=%u\nbuffers_per_volume=%u\ndepth=%u\nfile_path=%s\nheight=%u\nwait_time=%u\nwidth=%u\n",
		        iniEscapedKey("Virtual OCT System").c_str(), vsys->bitDepth, vsys->buffersFromFile, vsys->buffersPerVolume, vsys->depth, S(vsysFilePath).c_str(), vsys->height,
		        vsys->waitTimeUs, vsys->width);
		fprintf(f, "copy_file_to_ram=%s\nbscan_offset=%u\nsync_with_processing=%s\n", B(vsys->copyFileToRam), vsys->bscanOffset, B(vsys->syncWithProcessing));
	}
	fclose(f);
	return OCTPIPE_OK;
}

// OctAlgorithmParametersManager::loadCurveFromFile (octalgorithmparametersmanager.cpp:12-30): the first line is dropped, every further
// line -- QTextStream::readLine: ended by \n or \r\n, the last one by the end of the file -- gives one value, field 1 of a split at
// ';' through QString::toFloat (no second field, text, a decimal comma, a hexadecimal form, a value outside the float range: 0)
int octhost_load_curve_csv(const char* path, float* out, unsigned capacity, unsigned* count) {
	if (!path || !count) return OCTPIPE_ERR_INVALID_ARGUMENT;
	*count = 0;
	std::ifstream f(path, std::ios::binary);
	if (!f) return OCTPIPE_ERR_INVALID_ARGUMENT;  // the reference returns an empty curve (:17-20)
	std::string line;
	std::getline(f, line);  // header line is skipped unconditionally (:23)
	unsigned n = 0;
	while (std::getline(f, line)) {
		if (!line.empty() && line.back() == '\r') line.pop_back();
		// QString::section(";", 1, 1): the text between the first and the second ';' (empty -> 0.0f)
		size_t a = line.find(';');
		std::string field;
		if (a != std::string::npos) {
			size_t b = line.find(';', a + 1);
			field = line.substr(a + 1, b == std::string::npos ? std::string::npos : b - a - 1);
		}
		if (out && n < capacity) out[n] = qtToFloat(field);
		++n;
	}
	*count = n;
	return OCTPIPE_OK;
}

// saveCurveToFile (:32-45): "Sample Number;Sample Value", then "<i>;<value>" with QTextStream's default float format
int octhost_save_curve_csv(const char* path, const float* curve, unsigned count) {
	if (!path || (!curve && count)) return OCTPIPE_ERR_INVALID_ARGUMENT;
	FILE* f = fopen(path, "wb");
	if (!f) return OCTPIPE_ERR_INVALID_ARGUMENT;
	fprintf(f, "Sample Number;Sample Value\n");  // :35
	for (unsigned i = 0; i < count; ++i) fprintf(f, "%u;%s\n", i, qtStreamFloat(curve[i]).c_str());
	fclose(f);
	return OCTPIPE_OK;
}

}  // extern "C"
