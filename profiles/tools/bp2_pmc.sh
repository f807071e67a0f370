# PMC passes over the stand-alone bit-plane prototype (main configuration only):  bash profiles/tools/bp2_pmc.sh
repo=$(pwd); export TMPDIR=/tmp; cd /tmp
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INST_LEVEL_VMEM" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_WAVES"; do
  i=$((i+1)); rm -rf /tmp/pp$i
  timeout 300 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d /tmp/pp$i -- $repo/profiles/tools/bp2.bin 125000 10000 1 > /tmp/pp$i.log 2>&1
done
cd $repo
python profiles/summarise_pmc.py a=/tmp/pp1 b=/tmp/pp2 c=/tmp/pp3 d=/tmp/pp4 e=/tmp/pp5 f=/tmp/pp6 | grep -E "k_bp2"
