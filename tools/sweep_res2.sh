#!/bin/bash
# automatic choice (resident second generation / first generation / streaming) vs streaming forced, over instance sizes: tools/kbench.py per line
for prec in float double; do
for rows in 50000 75000 100000 150000 200000; do
  vars=$((rows*2))
  for res in 0 1; do
      r=$(timeout 120 python tools/kbench.py --mt 1 --precision $prec --vars $vars --rows $rows --res $res --iters 300 2>/dev/null | tail -2 | tr '\n' ' ')
      echo "$prec rows=$rows res=$res: $r"
  done
done
done
