R=$(QPGPU_WITNESS_DEBUG_P2ROWS=1 python3 /root/repo/tools/scratch_hint_exp.py 2>&1 | grep "^P2ROWS" | head -1 | sed 's/P2ROWS//')
echo "rows: $R" | cut -c1-200
echo "--- default plan"; P2ROWS="$R" python3 /root/repo/tools/scratch_hint_exp.py 2>&1 | grep -v "^P2ROWS" | tail -6
echo "--- assignment-aware plan"; QPGPU_WITNESS_ASSIGNED_PLAN=1 P2ROWS="$R" python3 /root/repo/tools/scratch_hint_exp.py 2>&1 | grep -v "^P2ROWS" | tail -6
