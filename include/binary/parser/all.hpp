// binary/parser/all.hpp — one include for the parser side of the drop-in (the reference has an umbrella header of the
// same path, library/include/binary/parser/all.hpp).
#ifndef BINARY_AMD_PARSER_ALL_HPP_
#define BINARY_AMD_PARSER_ALL_HPP_

#include <binary/parser/vcf.hpp>

#endif  // BINARY_AMD_PARSER_ALL_HPP_
