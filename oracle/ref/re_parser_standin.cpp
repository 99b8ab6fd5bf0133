/*
 * TEST INFRASTRUCTURE (oracle/_ref build only) -- not part of the product path.
 *
 * Stand-in for the bison output of /root/reference/pire/re_parser.y.
 *
 * The reference library's only generated translation unit is re_parser.cpp
 * (pire/Makefile.am lists it among libpire_la_SOURCES; it is produced from
 * re_parser.y by bison, which this image does not have).  It provides exactly
 * one symbol the rest of the library needs:
 *
 *     int Pire::Impl::yre_parse(Pire::Lexer&)          (re_parser.y:264-275)
 *
 * called from Lexer::Parse() (re_lexer.cpp:379-387).  This file implements
 * that symbol as a recursive-descent parser over the same token stream
 * (Lexer::Lex(), re_lexer.cpp:143-186) with the same semantic actions, rule by
 * rule.  It is cold-path code: it decides WHICH Fsm gets built from a pattern
 * string and has no part in how a compiled scanner table is walked.
 *
 * Acceptance gate: with this file the reference's own tests/pire_ut.cpp and
 * tests/easy_ut.cpp build and pass (see oracle/Makefile target `reftest`).
 *
 * Grammar (re_parser.y:82-159), LALR there, LL(1) here:
 *
 *   regexp        : alternative                        <end of input>
 *   alternative   : conjunction ( '|' conjunction )*
 *   conjunction   : negation ( YRE_AND negation )*
 *   negation      : [ YRE_NOT ] concatenation
 *   concatenation : iteration*
 *   iteration     : term [ YRE_COUNT ]
 *   term          : YRE_LETTERS | YRE_DOT | '^' | '$' | '(' alternative ')'
 */

#include <memory>
#include <stdexcept>

#include "fsm.h"
#include "re_lexer.h"
#include "re_parser.h"
#include "any.h"
#include "stub/stl.h"

namespace {

using Pire::Any;
using Pire::Encoding;
using Pire::Fsm;
using Pire::Lexer;
using Pire::Term;
using Pire::TVector;
using Pire::ystring;

typedef std::unique_ptr<Any> Value;

struct SyntaxError {};

/* Appends a literal character set to `a` -- semantic helper, re_parser.y:190-213. */
void AppendRange(const Encoding& encoding, Fsm& a, const Term::CharacterRange& range)
{
	TVector<ystring> local;
	for (auto&& ucs4 : range.first) {
		ystring bytes;
		bool representable = true;
		for (auto&& ch : ucs4) {
			ystring piece = encoding.ToLocal(ch);
			if (piece.empty()) {
				representable = false;
				break;
			}
			bytes += piece;
		}
		if (representable && !bytes.empty())
			local.push_back(bytes);
	}
	if (local.empty())
		a = Fsm::MakeFalse();      // nothing in the set exists in this encoding
	else
		a.AppendStrings(local);
}

/* Turns a token value into an Fsm in place -- re_parser.y:215-243. */
Fsm& ToFsm(const Encoding& encoding, Any* value)
{
	if (value->IsA<Fsm>())
		return value->As<Fsm>();

	Any holder = Fsm();
	Fsm& a = holder.As<Fsm>();

	if (value->IsA<Term::DotTag>()) {
		encoding.AppendDot(a);
	} else if (value->IsA<Term::BeginTag>()) {
		a.AppendSpecial(Pire::BeginMark);
	} else if (value->IsA<Term::EndTag>()) {
		a.AppendSpecial(Pire::EndMark);
	} else {
		Term::CharacterRange range = value->As<Term::CharacterRange>();
		AppendRange(encoding, a, range);
		if (range.second) {
			// negated set [^...]: (set | ~dot) complemented, dead ends dropped
			Fsm anyChar;
			encoding.AppendDot(anyChar);
			anyChar.Complement();
			a |= anyChar;
			a.Complement();
			a.RemoveDeadEnds();
		}
	}
	value->Swap(holder);
	return a;
}

class Parser {
public:
	explicit Parser(Lexer& lexer): m_lexer(lexer), m_token(0), m_have(false) {}

	/* regexp : alternative  -- re_parser.y:82-90 */
	void ParseRegexp()
	{
		Value top = Alternative();
		if (Tok() != 0)
			throw SyntaxError();
		ToFsm(Enc(), top.get());
		Pire::DoSwap(m_lexer.Retval(), *top);
	}

private:
	const Encoding& Enc() const { return m_lexer.Encoding(); }

	/* Token fetch with the value boxed as bison's yylex does -- re_parser.y:163-176.
	 *
	 * The lookahead is fetched LAZILY, only when a decision needs it.  This mirrors the generated parser, which
	 * performs default reductions without reading a lookahead token: after shifting ')' (or a YRE_COUNT) the only
	 * possible action is the reduction whose semantic action calls Lexer::Parenthesized (re_parser.y:148, 158), so
	 * that call happens BEFORE the next token is lexed.  Lexer features keep state between their Lex() and
	 * Parenthesized() calls (extra/capture.cpp:40-87 counts '(' tokens), so the interleaving is observable:
	 * tests/capture_ut.cpp (SlowCapturing, a group directly followed by '(') fails with an eager lookahead. */
	int Tok()
	{
		if (!m_have) {
			Term t = m_lexer.Lex();
			m_value.reset(t.Value().Empty() ? nullptr : new Any(t.Value()));
			m_token = t.Type();
			m_have = true;
		}
		return m_token;
	}

	/* Consume the current token (its value is returned); the next one is NOT fetched yet. */
	Value Shift()
	{
		Tok();
		m_have = false;
		return std::move(m_value);
	}

	bool AtTerm()
	{
		const int t = Tok();
		return t == YRE_LETTERS || t == YRE_DOT || t == '^' || t == '$' || t == '(';
	}

	/* alternative : conjunction | alternative '|' conjunction  -- re_parser.y:92-95 */
	Value Alternative()
	{
		Value lhs = Conjunction();
		while (Tok() == '|') {
			Shift();
			Value rhs = Conjunction();
			Fsm& l = ToFsm(Enc(), lhs.get());
			l |= ToFsm(Enc(), rhs.get());
		}
		return lhs;
	}

	/* conjunction : negation | conjunction YRE_AND negation  -- re_parser.y:97-100 */
	Value Conjunction()
	{
		Value lhs = Negation();
		while (Tok() == YRE_AND) {
			Shift();
			Value rhs = Negation();
			Fsm& l = ToFsm(Enc(), lhs.get());
			l &= ToFsm(Enc(), rhs.get());
		}
		return lhs;
	}

	/* negation : concatenation | YRE_NOT concatenation  -- re_parser.y:102-105 */
	Value Negation()
	{
		if (Tok() != YRE_NOT)
			return Concatenation();
		Shift();
		Value body = Concatenation();
		ToFsm(Enc(), body.get()).Complement();
		return body;
	}

	/* concatenation : <empty> | concatenation iteration  -- re_parser.y:107-120 */
	Value Concatenation()
	{
		Value acc(new Any(Fsm()));
		while (AtTerm()) {
			Value item = Iteration();
			Fsm& a = ToFsm(Enc(), acc.get());
			if (item->IsA<Term::CharacterRange>() && !item->As<Term::CharacterRange>().second)
				AppendRange(Enc(), a, item->As<Term::CharacterRange>());
			else if (item->IsA<Term::DotTag>())
				Enc().AppendDot(a);
			else
				a += ToFsm(Enc(), item.get());
		}
		return acc;
	}

	/* iteration : term | term YRE_COUNT  -- re_parser.y:122-151 */
	Value Iteration()
	{
		Value base = ParseTerm();
		if (Tok() != YRE_COUNT)
			return base;
		Value count = Shift();

		Fsm& orig = ToFsm(Enc(), base.get());
		Value result(new Any(orig));
		Fsm& cur = result->As<Fsm>();
		const Term::RepetitionCount& rep = count->As<Term::RepetitionCount>();
		const int lo = rep.first, hi = rep.second;

		if (lo == 0 && hi == 1) {
			Fsm empty;
			cur |= empty;
		} else if (lo == 0 && hi == Pire::Consts::Inf) {
			cur.Iterate();
		} else if (lo == 1 && hi == Pire::Consts::Inf) {
			cur += *cur;
		} else {
			cur *= lo;
			if (hi == Pire::Consts::Inf)
				cur += *orig;
			else if (hi != lo)
				cur += (orig | Fsm()) * (hi - lo);
		}
		m_lexer.Parenthesized(result->As<Fsm>());
		return result;
	}

	/* term : LETTERS | DOT | '^' | '$' | '(' alternative ')'  -- re_parser.y:153-159 */
	Value ParseTerm()
	{
		if (Tok() == '(') {
			Shift();
			Value inner = Alternative();
			if (Tok() != ')')
				throw SyntaxError();
			Shift();
			// as in the grammar action, $2 is used as an Fsm here
			m_lexer.Parenthesized(inner->As<Fsm>());
			return inner;
		}
		if (!AtTerm())
			throw SyntaxError();
		return Shift();
	}

	Lexer& m_lexer;
	int    m_token;
	bool   m_have;     // m_token / m_value hold an unconsumed lookahead
	Value  m_value;
};

} // namespace

namespace Pire {
namespace Impl {

/* Same contract as the generated one (re_parser.y:264-275): 0 on success, non-zero on a
 * syntax error (Lexer::Parse then throws "Syntax error in regexp"); lexer errors propagate
 * as Pire::Error exactly as Lexer::Lex throws them. */
int yre_parse(Pire::Lexer& rlex)
{
	try {
		Parser(rlex).ParseRegexp();
	} catch (const SyntaxError&) {
		return 1;
	}
	if (!rlex.ErrMsg().empty())
		throw Error(rlex.ErrMsg());
	return 0;
}

}
}
