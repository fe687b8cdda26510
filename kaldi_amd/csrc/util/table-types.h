// Table typedefs used by the nnet1 tools (reference util/table-types.h:34-57).
#ifndef KALDI_AMD_UTIL_TABLE_TYPES_H_
#define KALDI_AMD_UTIL_TABLE_TYPES_H_
#include "matrix/matrix-lib.h"
#include "util/kaldi-table.h"

namespace kaldi {
typedef TableWriter<KaldiObjectHolder<Matrix<BaseFloat> > > BaseFloatMatrixWriter;
typedef SequentialTableReader<KaldiObjectHolder<Matrix<BaseFloat> > > SequentialBaseFloatMatrixReader;
typedef RandomAccessTableReader<KaldiObjectHolder<Matrix<BaseFloat> > > RandomAccessBaseFloatMatrixReader;
typedef PrefetchingTableReader<KaldiObjectHolder<Matrix<BaseFloat> > > PrefetchingBaseFloatMatrixReader;
typedef AsyncTableWriter<KaldiObjectHolder<Matrix<BaseFloat> > > AsyncBaseFloatMatrixWriter;
typedef TableWriter<KaldiObjectHolder<Vector<BaseFloat> > > BaseFloatVectorWriter;
typedef SequentialTableReader<KaldiObjectHolder<Vector<BaseFloat> > > SequentialBaseFloatVectorReader;
typedef RandomAccessTableReader<KaldiObjectHolder<Vector<BaseFloat> > > RandomAccessBaseFloatVectorReader;
typedef TableWriter<BasicVectorHolder<int32> > Int32VectorWriter;
typedef SequentialTableReader<BasicVectorHolder<int32> > SequentialInt32VectorReader;
typedef RandomAccessTableReader<BasicVectorHolder<int32> > RandomAccessInt32VectorReader;
}  // namespace kaldi
#endif
