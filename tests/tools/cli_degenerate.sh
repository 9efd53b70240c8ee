#!/bin/bash
# degenerate inputs through both programs (GPU box, needs oracle/_ref/peng_motif_ref): exit codes and stdout must agree
T=$(mktemp -d); R=oracle/_ref/peng_motif_ref; H=peng-motif_amd/host/peng_motif
printf "" > $T/empty.fa
printf ">a\n>b\n" > $T/headers_only.fa
printf ">a\nACGT\n>b\nAC\n" > $T/short.fa
printf ">a\nNNNNNNNNNNNNNNNNNNNNNNNNNNNNNN\n>b\nNNNNNNNNNNNNNNNNN\n" > $T/alln.fa
printf ">a\nACGTACGTACGTACGTACGTACGTACGTACGTACGTACGT\n" > $T/one.fa
printf ">a\nAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAA\n>b\nAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAA\n" > $T/polya.fa
printf ">a\nACGTTGCAAGCTAGCTAGGATCGATCGATTAGCTAGCTAGCTAGGGATCGA" > $T/no_newline.fa
printf ">a\r\nACGTTGCAAGCTAGCTAGGATCGATCGATTAGC\r\n>b\r\nACGTTGCAAGCTAGCTAGGATCGATCGATTAGC\r\n" > $T/crlf.fa
for f in empty headers_only short alln one polya no_newline crlf; do
  for w in 4 8; do
    $R $T/$f.fa -w $w -o $T/r.meme > $T/r.out 2> $T/r.err; rr=$?
    $H $T/$f.fa -w $w -o $T/h.meme > $T/h.out 2> $T/h.err; hr=$?
    so=same; cmp -s $T/r.out $T/h.out || so=DIFF
    sm=same; if [ -f $T/r.meme ] || [ -f $T/h.meme ]; then cmp -s $T/r.meme $T/h.meme || sm=DIFF; fi
    echo "$f w=$w rc ref=$rr here=$hr stdout=$so meme=$sm"
    rm -f $T/r.meme $T/h.meme
  done
done
rm -rf $T
