set -x
OUT=gpurun_out/r03u
mkdir -p $OUT
for i in 1 2 3 4; do
  ( timeout 900 python -m pytest tests/test_drivers_gpu.py -m gpu -x -q -k "train_retriever" ) > $OUT/pytest_$i.log 2>&1
  tail -1 $OUT/pytest_$i.log
done
python train_retriever.py --synthetic 4,6,24,40 --retriever_layers 2 --optim adamw --scheduler linear --lr 1e-3 --weight_decay 0.01 --dropout 0.1 --epochs 3 --checkpoint_dir /tmp/ck_ret --name ret --per_gpu_batch_size 4 --steps 40 2>&1 | grep -i "train:\|eval:" | head -20
