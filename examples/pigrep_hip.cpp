// pigrep on the MI355X: the reference's sample grep (samples/pigrep/pigrep.cpp) with its one-line-at-a-time
//     if (Pire::Runner(sc).Begin().Run(line).End()) print(line)
// loop (pigrep.cpp:38-45) replaced by ONE batched call on the GPU.  Same command line, same output.
//
//   pigrep_hip [-i] [-u] [-x] [-e pattern | pattern] [file [file2...]]
//
// Everything before the scan (lexer, features, Surround, Compile) is the reference library, unchanged; the scan goes
// through include/pire_hip/batch_runner.hpp.  tests/test_examples.py builds the reference's own pigrep next to this
// one and compares their outputs.
#include <cstring>
#include <fstream>
#include <iostream>
#include <iterator>
#include <stdexcept>
#include <string>
#include <vector>

#include <pire/pire.h>
#include <pire_hip/batch_runner.hpp>

namespace {

// All lines of the stream in one buffer + their offsets (getline semantics: the newline is not part of the line, a
// trailing fragment without newline is a line, an empty stream has no lines).
void ReadLines(std::istream& in, std::string* text, std::vector<uint64_t>* offsets)
{
	const std::string raw((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
	text->clear();
	offsets->assign(1, 0);
	size_t pos = 0;
	while (pos < raw.size()) {
		size_t nl = raw.find('\n', pos);
		if (nl == std::string::npos)
			nl = raw.size();
		text->append(raw, pos, nl - pos);
		offsets->push_back(text->size());
		pos = nl + 1;
	}
}

void GrepStream(std::istream& in, const Pire::Hip::Table<Pire::Scanner>& table, const std::string& prefix)
{
	std::string text;
	std::vector<uint64_t> offsets;
	ReadLines(in, &text, &offsets);
	const size_t n = offsets.size() - 1;
	if (!n)
		return;
	Pire::Hip::BatchRunner<Pire::Scanner> run(table);
	const std::vector<char>& hit = run.Begin().Run(text.data(), offsets.data(), n).End().Finals();
	for (size_t i = 0; i < n; ++i)
		if (hit[i])
			std::cout << prefix << text.substr(offsets[i], offsets[i + 1] - offsets[i]) << std::endl;
}

void Usage()
{
	std::cerr << "pigrep_hip: print the lines a Pire regexp matches, scanning on the GPU\n"
	             "  pigrep_hip [-i] [-u] [-x] (-e PATTERN | PATTERN) [FILE...]\n"
	             "    -i  ignore case            -u  pattern and text are UTF-8\n"
	             "    -x  allow re1&re2 and ~re  -e  next argument is the pattern, even if it starts with '-'\n"
	             "  no FILE (or '-'): standard input; several FILEs: lines are prefixed with the file name\n";
	exit(1);
}

}  // namespace

int main(int argc, char** argv)
{
	try {
		Pire::Lexer lexer;
		std::string pattern;
		bool havePattern = false;
		int arg = 1;
		for (; arg < argc; ++arg) {
			const std::string a = argv[arg];
			if (a == "-i") {
				lexer.AddFeature(Pire::Features::CaseInsensitive());
			} else if (a == "-u") {
				lexer.SetEncoding(Pire::Encodings::Utf8());
			} else if (a == "-x") {
				lexer.AddFeature(Pire::Features::AndNotSupport());
			} else if (a == "-e" && arg + 1 < argc && !havePattern) {
				pattern = argv[++arg];
				havePattern = true;
			} else if (a.size() > 1 && a[0] == '-') {
				Usage();
			} else if (!havePattern) {
				pattern = a;
				havePattern = true;
			} else {
				break;
			}
		}
		if (!havePattern)
			Usage();

		// pigrep.cpp:88-94: decode the pattern with the lexer's encoding, parse, Surround, compile
		Pire::TVector<Pire::wchar32> ucs4;
		lexer.Encoding().FromLocal(pattern.data(), pattern.data() + pattern.size(), std::back_inserter(ucs4));
		lexer.Assign(ucs4.begin(), ucs4.end());
		Pire::Scanner sc = lexer.Parse().Surround().Compile<Pire::Scanner>();
		Pire::Hip::Table<Pire::Scanner> table(sc);   // one device table for every file

		std::ios_base::sync_with_stdio(false);
		if (arg >= argc) {
			GrepStream(std::cin, table, "");
		} else {
			// pigrep.cpp:93-108: "-" is stdin; lines are prefixed with "name: " only when several files are given
			const bool many = argc - arg > 1;
			for (; arg < argc; ++arg) {
				const std::string name = argv[arg];
				if (name == "-") {
					GrepStream(std::cin, table, many ? "(stdin): " : "");
					continue;
				}
				std::ifstream f(name.c_str());
				if (!f)
					throw std::runtime_error("cannot open file " + name);
				GrepStream(f, table, many ? name + ": " : std::string());
			}
		}
		return 0;
	} catch (const std::exception& e) {
		std::cerr << "pigrep_hip: " << e.what() << std::endl;
		return 1;
	}
}
