// bam.h -- unaligned BAM / SAM input, decoded to FASTQ text in memory.
// The reference reads these through htslib (read_bam, src/TGSFilter.cpp:984-1040, :1872-1917): per
// record it takes the query name, the bases through Base[16] = {0,'A','C',0,'G',0,0,0,'T',0,...,'N'}
// applied to the 4-bit codes (:31, :1898) -- so IUPAC codes become NUL bytes, SAM text is normalised
// to upper case -- and the qualities as value + 33 (:1899); flags, strand and tags are ignored.
// htslib is not linked here: BGZF is a series of gzip members (zlib), the BAM record layout and
// the SAM text -> 4-bit code table are restated from the SAM specification.
#pragma once
#include <cstddef>
#include <memory>
#include <string>
#include <vector>

#include "textsource.h"

namespace host {

// The same decoders as streams (textsource.h): bounded memory whatever the size of the file.
std::unique_ptr<ByteStream> make_gz_bytes(const char* data, size_t size);                 // any series of gzip members
std::unique_ptr<TextSource> make_bam_text(std::unique_ptr<ByteStream> bytes);             // decompressed BAM -> FASTQ text
std::unique_ptr<TextSource> make_sam_text(std::unique_ptr<ByteStream> bytes);             // SAM text -> FASTQ text

// true if `data` starts like BGZF/gzip or like BAM/SAM is irrelevant here: the caller decides by file name
// (file_type() == 2, src/TGSFilter.cpp:833); the content decides BAM vs SAM, as hts_open does.
// Appends "@name\nSEQ\n+\nQUAL\n" per record to `text`.  Returns false (message in `err`) on malformed input.
bool decode_sam_or_bam(const char* data, size_t size, std::vector<char>& text, std::string& err);

// gzip -> bytes: any series of gzip members (zlib); BGZF members (bgzip, BAM) are located from their 'BC'
// fields and inflated side by side on several threads.
bool inflate_gzip(const char* data, size_t size, std::vector<char>& out, std::string& err);

}  // namespace host
