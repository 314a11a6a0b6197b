import sys, time, json; sys.path.insert(0,'/root/repo')
import torch
from robustart_amd.train import cls_solver as S
class A: pass
B=int(sys.argv[1]) if len(sys.argv)>1 else 256
for eng, adv in (('hip', None), ('torch', None), ('hip', {'eps':'4/255','steps':3,'rel_stepsize':0.4}), ('torch', {'eps':'4/255','steps':3,'rel_stepsize':0.4})):
    a=A(); a.engine='hip'; a.train_engine=eng; a.max_iter=6
    cfg={'model':{'type':'resnet50_official'},'data':{'fake_size':B*2,'batch_size':B,'input_size':224},'label_smooth':0.1,
         'ema':{'enable':True,'kwargs':{'decay':0.9999}},'max_iter':6,'saver':{'print_freq':100},
         'optimizer':{'type':'SGD','kwargs':{'nesterov':True,'momentum':0.9,'weight_decay':1e-4}},
         'lr_scheduler':{'kwargs':{'base_lr':0.1,'warmup_lr':0.4}}}
    if adv: cfg['adv_train']=adv
    # time: run once to warm (6 its), then again timed
    S.train(cfg,a,0,1,torch.device('cuda'))
    torch.cuda.synchronize(); t0=time.time()
    loss,_=S.train(cfg,a,0,1,torch.device('cuda'))
    torch.cuda.synchronize(); dt=(time.time()-t0)/6
    print('train_engine=%s adv=%s: %.1f ms/step (%.0f img/s)  loss %.3f  (includes engine construction amortised over 6 its)' % (eng, bool(adv), dt*1e3, B/dt, loss))
