#!/bin/bash
# malformed and unusual command lines through both programs (GPU box, needs oracle/_ref/peng_motif_ref): exit code, stdout,
# stderr and (where written) MEME / JSON must agree.  A crash of the reference (an uncaught std::stoi exception: SIGABRT)
# counts as "nonzero": the mirror must not succeed where the reference dies, and need not die the same way.
T=$(mktemp -d); R=$PWD/oracle/_ref/peng_motif_ref; H=$PWD/peng-motif_amd/host/peng_motif; F=$PWD/tests/golden/MafK_100seqs.fasta
LOG=$PWD/gpurun_out/r05_cli_args_differential.log; mkdir -p $PWD/gpurun_out; : > $LOG
cd $T
run() {  # name, then the arguments
  local name=$1; shift
  rm -f r.meme h.meme r.json h.json
  timeout -k 5 120 "$R" "${@//OUT/r}" > r.out 2> r.err0; rr=$?
  timeout -k 5 120 "$H" "${@//OUT/h}" > h.out 2> h.err0; hr=$?
  # (the logger's time stamps; "terminate called ..." of an uncaught exception names the mangled type on both sides)
  sed 's/^- [0-9:.]* /- TIME /' r.err0 > r.err; sed 's/^- [0-9:.]* /- TIME /' h.err0 > h.err
  local so=same se=same sm=same sj=same rc=same
  cmp -s r.out h.out || so=DIFF
  cmp -s r.err h.err || se=differs
  if [ -f r.meme ] || [ -f h.meme ]; then cmp -s r.meme h.meme || sm=DIFF; fi
  if [ -f r.json ] || [ -f h.json ]; then cmp -s r.json h.json || sj=DIFF; fi
  if [ $rr -ne $hr ]; then if [ $rr -ge 128 ] && [ $hr -ne 0 ]; then rc="ref-crashed($rr)/here=$hr"; else rc="DIFF($rr/$hr)"; fi; fi
  echo "$name: rc=$rc($rr) stdout=$so stderr=$se meme=$sm json=$sj" | tee -a $LOG
}
run no_args
run help -h
run help_after_file $F -h
run version -version
run version2 $F --version
run unknown_option $F --frobnicate
run unknown_short $F -q
run w_missing $F -w
run w_odd $F -w 7
run w_zero $F -w 0 -o OUT.meme
run w_negative $F -w -2
run w_text $F -w abc
run w_float $F -w 8.7 -o OUT.meme
# (-w 16 and above: the reference allocates 4^16 tables and runs for hours; the mirror ends with the reference's "too long" lines, tests/test_gpu_cli.py)
run t_text $F -w 8 -t abc
run t_negative $F -w 8 -t -3 -o OUT.meme --max-optimized-patterns 5
run strand_bad $F -w 8 --strand MINUS
run strand_lower $F -w 8 --strand plus
run score_bad $F -w 8 --optimization_score FOO
run bg_order_3 $F -w 8 --bg-model-order 3
run bg_order_neg $F -w 8 --bg-model-order -1
run bg_order_ge_w $F -w 4 --bg-model-order 2 -o OUT.meme
run threads_zero $F -w 8 --threads 0 -o OUT.meme
run threads_text $F -w 8 --threads many
run count_thr_0 $F -w 6 --count-threshold 0 -o OUT.meme --max-optimized-patterns 8
run em_it_0 $F -w 8 --em-max-iterations 0 -o OUT.meme
run em_it_neg $F -w 8 --em-max-iterations -5 -o OUT.meme
run em_thr_neg $F -w 8 --em-threshold -1 --em-max-iterations 4 -o OUT.meme
run sat_zero $F -w 8 -a 0 -o OUT.meme
run sat_neg $F -w 8 -a -10 -o OUT.meme
run pseudo_0 $F -w 8 --pseudo-counts 0 -o OUT.meme
run bitfactor_0 $F -w 8 -b 0 -o OUT.meme
run bitfactor_2 $F -w 8 -b 2 -o OUT.meme
run merged_len_small $F -w 8 --max_merged_length 4 -o OUT.meme
run max_opt_0 $F -w 8 --max-optimized-patterns 0 -o OUT.meme
run min_proc_big $F -w 8 --minimum-processed-patterns 100000 -o OUT.meme
run enrich_neg $F -w 8 --optimization_score ENRICHMENT --enrich_pseudocount_factor -1 -o OUT.meme
run out_unwritable $F -w 8 -o /nonexistent_dir/x.meme
run json_unwritable $F -w 8 -j /nonexistent_dir/x.json
run out_and_json $F -w 8 -o OUT.meme -j OUT.json
run option_twice $F -w 6 -w 8 -o OUT.meme
run file_twice $F $F -w 8
run bg_missing $F -w 8 --background-sequences /nonexistent.fa
run bg_same $F -w 8 --background-sequences $F -o OUT.meme
run value_missing_at_end $F -w 8 -t
cd /; rm -rf $T
