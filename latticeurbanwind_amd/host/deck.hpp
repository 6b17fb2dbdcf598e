// deck.hpp -- reader for LUW deck files (*.luw / *.luwpf / *.luwdg).
// The FORMAT is the contract (shared by the reference's C++ solver, its Python tools and its GUI: FX/setup.cpp:61-178,
// core/deck_io.py:87-423): one `key = value` per line; `//` starts a comment unless it sits inside '...' or "..."; keys are
// case-insensitive, blanks and dashes inside a key read as `_`, a few legacy spellings alias to current names; the last
// assignment of a key wins; values are kept as text (quotes included) and interpreted by whoever asks for them.
// This reader scans each line once with a small state machine and keeps the entries in a sorted table.
#pragma once
#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdlib>
#include <istream>
#include <string>
#include <utility>
#include <vector>

namespace luw_host {

class Deck {
public:
	typedef std::pair<std::string, std::string> Entry;

	static std::string strip(const std::string& s) {
		size_t a = 0u, b = s.size();
		while(a<b&&is_blank(s[a])) a++;
		while(b>a&&is_blank(s[b-1u])) b--;
		return s.substr(a, b-a);
	}
	// value without one pair of matching outer quotes
	static std::string text(const std::string& raw) {
		const std::string v = strip(raw);
		if(v.size()>=2u&&(v.front()=='"'||v.front()=='\'')&&v.back()==v.front()) return strip(v.substr(1u, v.size()-2u));
		return v;
	}
	// on / off words in any case, or any finite number (non-zero = on); false when the text is neither
	static bool flag(const std::string& raw, bool& value) {
		std::string w = text(raw);
		for(char& ch : w) ch = (char)std::tolower((unsigned char)ch);
		if(w.empty()) return false;
		static const char* const on_words[] = { "1", "enable", "enabled", "on", "t", "true", "y", "yes" };
		static const char* const off_words[] = { "0", "disable", "disabled", "f", "false", "n", "no", "off" };
		for(const char* k : on_words) if(w==k) { value = true; return true; }
		for(const char* k : off_words) if(w==k) { value = false; return true; }
		char* end = nullptr;
		const double number = std::strtod(w.c_str(), &end);
		if(end==w.c_str()||*end!='\0'||!std::isfinite(number)) return false;
		value = number!=0.0;
		return true;
	}

	void load(std::istream& in) {
		std::string line;
		while(std::getline(in, line)) {
			std::string key, value;
			if(split_line(line, key, value)) assign(canonical_key(key), value);
		}
	}
	const std::vector<Entry>& entries() const { return table; }
	const std::string* find(const std::string& key) const {
		const auto it = std::lower_bound(table.begin(), table.end(), key, [](const Entry& e, const std::string& k) { return e.first<k; });
		return (it!=table.end()&&it->first==key) ? &it->second : nullptr;
	}

private:
	std::vector<Entry> table; // sorted by key

	static bool is_blank(const char ch) { return ch==' '||ch=='\t'||ch=='\r'||ch=='\n'; }
	void assign(const std::string& key, const std::string& value) {
		if(key.empty()) return;
		const auto it = std::lower_bound(table.begin(), table.end(), key, [](const Entry& e, const std::string& k) { return e.first<k; });
		if(it!=table.end()&&it->first==key) it->second = value; else table.insert(it, Entry(key, value));
	}
	// One pass over the line: where does the comment start (a `//` outside quotes; a quote character only counts while the other
	// kind of quote is closed), and where is the first `=` in front of it?
	static bool split_line(const std::string& line, std::string& key, std::string& value) {
		enum { PLAIN, IN_SINGLE, IN_DOUBLE } state = PLAIN;
		size_t end = line.size(), eq = std::string::npos;
		for(size_t i=0u; i<line.size(); i++) {
			const char ch = line[i];
			if(ch=='='&&eq==std::string::npos) eq = i;
			if(i+1u==line.size()) break; // the last character can neither open a comment nor matter as a quote
			if(state==PLAIN) {
				if(ch=='\'') state = IN_SINGLE; else if(ch=='"') state = IN_DOUBLE;
				else if(ch=='/'&&line[i+1u]=='/') { end = i; break; }
			} else if((state==IN_SINGLE&&ch=='\'')||(state==IN_DOUBLE&&ch=='"')) state = PLAIN;
		}
		if(eq==std::string::npos||eq>=end) return false;
		key = line.substr(0u, eq);
		value = strip(line.substr(eq+1u, end-eq-1u));
		return true;
	}
	static std::string canonical_key(const std::string& raw) {
		std::string k;
		bool gap = false; // a run of blanks / dashes is pending
		for(const char ch : strip(raw)) {
			if(ch=='-'||std::isspace((unsigned char)ch)) { gap = true; continue; }
			if(gap&&!k.empty()) k.push_back('_');
			gap = false;
			k.push_back((char)std::tolower((unsigned char)ch));
		}
		size_t a = 0u, b = k.size();
		while(a<b&&k[a]=='_') a++;
		while(b>a&&k[b-1u]=='_') b--;
		k = k.substr(a, b-a);
		static const char* const renamed[][2] = { {"vk_inlet_aniso_scale", "vk_inlet_anisotropy"}, {"vk_inlet_anisotropy_scale", "vk_inlet_anisotropy"},
			{"vk_inlet_enable", "turb_inflow_enable"} };
		for(const auto& r : renamed) if(k==r[0]) return r[1];
		return k;
	}
};

} // namespace luw_host
