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
# (round 5: more of what real files contain)
printf ">a\nACGTTGCAAGCTAGCTAGGATCGATCGATTAGC\tACGT\n>b\nACGTTGCAAGCTAGCTAGGATCGATCGATTAGC\n" > $T/tab.fa
printf ">a\nACGTTGCAAG>CTAGCTAGGATCGATCGATTAGC\n>b\nACGTTGCAAGCTAGCTAGGATCGATCGATTAGC\n" > $T/gt_inside.fa
printf ">a\nACGTRYKMSWBDHVNACGTTGCAAGCTAGCTAGGATCGATCG\n>b\nACGTTGCAAGCTAGCTAGGATCGATCGATTAGCnnnnacgtacgtacgatcgatcagctacgatcg\n" > $T/iupac_letters.fa
printf ">a\nACGT1234ACGTTGCAAGCTAGCTAGG*ATCGA-TCGATTAGC\n>b\nACGTTGCAAGCTAGCTAGGATCGATCGATTAGC\n" > $T/digits_stars.fa
printf "\n\n>a\n\nACGTTGCAAGCTAGCTAGG\n\nATCGATCGATTAGC\n\n\n>b\nACGTTGCAAGCTAGCTAGGATCGATCGATTAGC\n\n" > $T/blank_lines.fa
printf "ACGTTGCAAGCTAGCTAGG\n>a\nACGTTGCAAGCTAGCTAGGATCGATCGATTAGC\n" > $T/no_first_header.fa
printf " >a\nACGTTGCAAGCTAGCTAGGATCGATCGATTAGC\n" > $T/leading_space.fa
(printf ">"; head -c 200000 /dev/zero | tr '\0' 'h'; printf "\nACGTTGCAAGCTAGCTAGGATCGATCGATTAGCACGTTGCAAGCTAGCTAGGATCGATCGATTAGC\n>b\nACGTTGCAAGCTAGCTAGGATCGATCGATTAGC\n") > $T/long_header.fa
printf ">\nACGTTGCAAGCTAGCTAGGATCGATCGATTAGC\n>\nACGTTGCAAGCTAGCTAGGATCGATCGATTAGC\n" > $T/empty_names.fa
mkdir $T/dir.fa
for f in empty headers_only short alln one polya no_newline crlf tab gt_inside iupac_letters digits_stars blank_lines no_first_header leading_space long_header empty_names dir missing; do
  for w in 4 8; do
    $R $T/$f.fa -w $w -o $T/r.meme > $T/r.out 2> $T/r.err; rr=$?
    $H $T/$f.fa -w $w -o $T/h.meme > $T/h.out 2> $T/h.err; hr=$?
    so=same; cmp -s $T/r.out $T/h.out || so=DIFF
    se=same; cmp -s $T/r.err $T/h.err || se=differs
    sm=same; if [ -f $T/r.meme ] || [ -f $T/h.meme ]; then cmp -s $T/r.meme $T/h.meme || sm=DIFF; fi
    echo "$f w=$w rc ref=$rr here=$hr stdout=$so meme=$sm stderr=$se"
    rm -f $T/r.meme $T/h.meme
  done
done
rm -rf $T
