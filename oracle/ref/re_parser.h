/*
 * TEST INFRASTRUCTURE (oracle/_ref build only) -- not part of the product path.
 *
 * Token ids that the reference lexer expects to find in the bison-generated
 * "re_parser.h" (used at /root/reference/pire/re_lexer.cpp:163-178; declared
 * as %term in /root/reference/pire/re_parser.y:69-73).  bison is not present
 * in this image, so the oracle build supplies this header and the
 * recursive-descent parser in re_parser_standin.cpp instead of bison output.
 * Values only need to be distinct and above the single-character tokens
 * ('(', ')', '|', '^', '$') that the grammar also uses.
 */
#pragma once

enum {
	YRE_LETTERS = 257,
	YRE_COUNT   = 258,
	YRE_DOT     = 259,
	YRE_AND     = 260,
	YRE_NOT     = 261
};
