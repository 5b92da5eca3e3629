#!/bin/bash
# schematic of an ISA range: tools/isa_sketch.sh file.s from to   (one line per MFMA with what follows it)
sed -n "$2,$3p" "$1" | grep -v "^\s*;" | awk '{print $1}' | awk '
/v_mfma/ {printf "\nMFMA |"; next}
/ds_read/ {printf " dsR"; next}
/ds_write/ {printf " dsW"; next}
/buffer_load_dwordx4/ {printf " BL"; next}
/buffer_store/ {printf " ST"; next}
/global_store/ {printf " GST"; next}
/global_load/ {printf " GL"; next}
/s_waitcnt/ {printf " WAIT"; next}
/s_barrier/ {printf " BAR"; next}
/s_cbranch/ {printf " BR"; next}
/s_nop/ {printf " nop"; next}
/^s_/ {printf " s"; next}
/^v_/ {printf " v"; next}
/^\./ {printf " LBL"; next}
{printf " ?%s", $1}
END {print ""}'
