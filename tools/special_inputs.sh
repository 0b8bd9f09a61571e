#!/bin/bash
# development aid: the CLI against the unmodified reference binary (oracle/_ref/cornetto) on inputs that are special rather than random — empty files, a lone
# newline or header, text without a header, blank records, a gzip file and a broken one; empty / one-line / mismatched / non-numeric / over-long bedgraph pairs, a
# track line, a missing file — stdout and exit status.  On the device path by default; CORNETTO_ACCEL=no bash tools/special_inputs.sh for the host path.
# (Not compared: `seq` on FASTA, where the reference prints "(null)", and binary garbage, on which it aborts.)
R=$PWD/oracle/_ref/cornetto
C=$PWD/cornetto_amd/cornetto
D=$(mktemp -d /dev/shm/special.XXXXXX) && cd $D || exit 2
printf "" > empty.fa; printf "\n" > nl.fa; printf ">" > gt.fa; printf "@" > at.fa; printf ">\n" > gtnl.fa; printf ">a\n" > hdr.fa; printf "ACGT\n" > nohdr.fa
printf ">a\n\n\n>b\nAC\n" > blank.fa; printf ">a\nACGT" | gzip > t.fa.gz; printf "\x1f\x8b" > badgz.fa.gz; printf "@r\nACGT\n+\nIIII\n@s\nAC\n+\nII\n" > r.fq
n=0; bad=0
for f in empty.fa nl.fa gt.fa at.fa gtnl.fa hdr.fa nohdr.fa blank.fa t.fa.gz badgz.fa.gz r.fq; do for sub in telofind sdust fa2bed "seq -m 0" "seq -m 3"; do
  case "$sub $f" in seq*.fa|seq*.fa.gz) continue;; esac
  r1=$(timeout 60 $R $sub $f 2>/dev/null >o1; echo $?); r2=$(timeout 60 $C $sub $f 2>/dev/null >o2; echo $?); n=$((n+1))
  if [ "$r1" != "$r2" ] || ! cmp -s o1 o2; then bad=$((bad+1)); echo "DIFF [$sub $f]: reference rc=$r1, here rc=$r2, stdout $(wc -c <o1) vs $(wc -c <o2) bytes"; fi
done; done
printf "" > e.bg; printf "c0\t0\t1\t5\n" > one.bg; printf "c0\t0\t1\t5\nc0\t1\t2\t6\n" > two.bg; printf "track type=bedGraph\nc0\t0\t1\t5\n" > trk.bg
printf "c0\t0\t1\t5\n\n\n" > tail.bg; printf "c0 0 1 5 c0 1 2 6\n" > oneline.bg; printf "c0\t0\t1\t5.5\n" > flt.bg; printf "c0\t0\t1\t99999999999\n" > big.bg
for pair in "e.bg e.bg" "one.bg one.bg" "two.bg one.bg" "one.bg two.bg" "trk.bg trk.bg" "tail.bg one.bg" "oneline.bg two.bg" "flt.bg flt.bg" "one.bg e.bg" "e.bg one.bg" "big.bg big.bg" "nonexist.bg one.bg" "one.bg nonexist.bg"; do
  set -- $pair
  for sub in noboringbits boringbits; do for o in "" "-w 1 -i 1 -m 0 -e 0"; do
    r1=$(timeout 60 $R $sub $1 -q $2 $o 2>/dev/null >o1; echo $?); r2=$(timeout 60 $C $sub $1 -q $2 $o 2>/dev/null >o2; echo $?); n=$((n+1))
    if [ "$r1" != "$r2" ] || ! cmp -s o1 o2; then bad=$((bad+1)); echo "DIFF [$sub $1 -q $2 $o]: reference rc=$r1, here rc=$r2, stdout $(wc -c <o1) vs $(wc -c <o2) bytes"; fi
  done; done
done
cd / && rm -rf $D
echo "special_inputs: $n command lines, $bad differences"
