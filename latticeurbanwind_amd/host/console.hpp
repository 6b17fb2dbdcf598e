// console.hpp -- console / log / text helpers of the deck driver: the reference's console rows, float text, parallel_for
// Part of the deck driver (luw_driver.cpp); included by it only, after lbm.hpp (namespace luw_host, std::string as string).
#pragma once

// ------------------------------------------------------------------------------------------------ console / text
static std::ofstream g_log;
static void println(const string& s = "") { std::cout << s << "\n"; std::cout.flush(); if(g_log.is_open()) { g_log << s << "\n"; g_log.flush(); } }
static const uint CONSOLE_WIDTH = 94u; // FX/utilities.hpp:9
// GUI protocol lines go to stdout only, never into the log (FX/utilities.hpp:3161-3178)
static const ProgressChannel g_progress([](const string& line) { std::cout << line << "\n"; std::cout.flush(); });
// the running row overwrites itself on the console (reprint, FX/info.cpp:273) and is not logged until it is final
static void reprint_row(const string& s) { std::cout << "\r" << s; std::cout.flush(); }
// LUW_DRIVER_TIMING=1: wall time of each phase of the run on stderr (profiling aid; console and log stay as the reference's)
static void phase_mark(const char* name) {
	static const bool on = [] { const char* e = std::getenv("LUW_DRIVER_TIMING"); return e&&e[0]=='1'; }();
	static auto last = std::chrono::steady_clock::now();
	if(!on) return;
	const auto now = std::chrono::steady_clock::now();
	std::fprintf(stderr, "[timing] %-28s %8.3f s\n", name, std::chrono::duration<double>(now-last).count());
	last = now;
}
// a field of n characters: text flush right / flush left (a longer text is kept whole), or centred (a longer text is cut to the field)
static string alignr(const uint n, const string& x) { return x.length()>=n ? x : string(n-x.length(), ' ')+x; }
static string alignl(const uint n, const string& x) { string field = x; if(field.length()<n) field.append(n-field.length(), ' '); return field; }
static string alignc(const uint n, const string& x) {
	if(x.length()>=n) return x.substr(0u, n);
	const size_t spare = n-x.length(), left = spare/2u;
	string field(left, ' ');
	field += x;
	field.append(spare-left, ' ');
	return field;
}
static string hr_plain() { return "|"+string(CONSOLE_WIDTH-2u, '-')+"|"; }
static void print_section_title(const string& t) { println(hr_plain()); println("|"+alignc(CONSOLE_WIDTH-2u, t)+"|"); println(hr_plain()); }
static void print_kv_row(const string& k, const string& v) { println("| "+k+" | "+v+" |"); }
static string to_string_u(ulong x) { string r; do { r = (char)(x%10ull+48ull)+r; x /= 10ull; } while(x); return r; }
// The reference prints floats with 9 significant digits -- "d.dddddddd[E<exp>]" -- and kernel constants and VTK headers travel
// through that text (FX/utilities.hpp:2603-2634,2741-2750), so the digits must be the reference's, float operation for float
// operation.  The decimal exponent comes from a binary ladder of powers of ten (each rung at most once, largest first); the nine
// digits from one truncation and one round-half-up.  One implementation for the whole project: luw_format_float9 in the library.
static string to_string_f(const float x) {
	char text[48];
	luw_format_float9(x, text, sizeof(text));
	return text;
}
// Fixed-point text "[-]iii.ddd" with `decimals` digits behind the point (at most MAXD), rounded half up by adding half a unit of the last digit in the
// value's own precision and truncating twice -- the float (or double) sequence of the reference's to_string(x, decimals), FX/utilities.hpp:2762-2783,
// whose digits end up in file names and console rows.
template<typename F, uint MAXD> static string fixed_decimals(F x, const uint decimals) {
	const uint digits = std::min(decimals, MAXD);
	const string sign = x<(F)0 ? "-" : "";
	if(x<(F)0) x = -x;
	if(std::isnan(x)) return sign+"NaN";
	if(std::isinf(x)) return sign+"Inf";
	const F power = std::pow((F)10, (F)digits);
	x += (F)0.5/power;
	const ulong whole = (ulong)x;
	ulong fraction = (ulong)((x-(F)whole)*power);
	if(sizeof(F)==4u) fraction = (ulong)(uint)fraction;
	string tail(digits, '0');
	for(uint d=digits; d>0u; d--) { tail[d-1u] = (char)('0'+fraction%10ull); fraction /= 10ull; }
	return sign+to_string_u(whole)+(decimals==0u ? "" : "."+tail);
}
static string to_string_fd(const float x, const uint decimals) { return fixed_decimals<float, 8u>(x, decimals); }
static string to_string_dd(const double x, const uint decimals) { return fixed_decimals<double, 16u>(x, decimals); }
static string fmtf(float v, int prec = 4) { std::ostringstream os; os << std::fixed; os.precision(prec); os << v; return os.str(); }
static string format_tag(float v) {
	string s = to_string_fd(v, 3u);
	if(s.find('.')!=string::npos) { while(!s.empty()&&s.back()=='0') s.pop_back(); if(!s.empty()&&s.back()=='.') s.pop_back(); }
	return s.empty() ? "0" : s;
}
static string now_str(const char* fmt = "%Y%m%d %H:%M:%S") {
	std::time_t tt = std::time(nullptr);
	std::tm tm{};
	localtime_r(&tt, &tm);
	char b[64];
	std::strftime(b, sizeof(b), fmt, &tm);
	return b;
}
[[noreturn]] static void fatal(const string& msg, const int code = -1) { println(msg); println(hr_plain()); std::exit(code); }

template<typename Fn> static void parallel_for(const ulong N, Fn fn) { // FX/utilities.hpp:64-97
	const uint threads = std::max(1u, std::min((uint)std::thread::hardware_concurrency(), 64u));
	std::vector<std::thread> pool;
	for(uint t=0u; t<threads; t++) pool.emplace_back([=]() { for(ulong n=N*(ulong)t/threads; n<N*(ulong)(t+1u)/threads; n++) fn(n); });
	for(auto& th : pool) th.join();
}
static inline float reverse_bytes(const float v) { uint32_t u; std::memcpy(&u, &v, 4); u = __builtin_bswap32(u); float r; std::memcpy(&r, &u, 4); return r; }

