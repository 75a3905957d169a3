"""PTO scripts the two parsers (the reference's pto_parser_type and the product's pto_script) are compared on:
the scripts of tests/test_frontend.py / tests/test_cli.py, hugin-dialect lines with back references, quoted
names with blanks, k (mask) lines, comment and blank lines, malformed items."""

CASES = {
    "frontend_full": '''# hugin project file
#hugin_ptoversion 2
p f2 w4000 h2000 v360  E11.5 R0 S100,3900,50,1950 n"TIFF_m c:LZW"
m i2

# image lines
#-hugin  cropFactor=1
i w3000 h2000 f0 v50 Ra0 Eev12 y0 p0 r0 a0.01 b-0.03 c0.02 d5 e-3 g0 t0 TrX0 TrY0 TrZ0 n"img0.jpg"
i w3000 h2000 f0 v=0 Eev13 y40.5 p-3.25 r1.5 a=0 b=0 c=0 d0 e0 g12 t-6 n"img1.jpg"
i w2000 h2000 f3 v180 Eev12.5 y-90 p0 r0 n"fish.tif"
''',
    "two_facets": 'p f2 w300 h150 v360 n"TIFF"\ni w200 h150 f0 v70 y10 p5 r2 a0.01 b-0.03 c0.02 d0 e0 n"a.tif"\ni w160 h160 f3 v170 y-100 p-20 r0 n"b.tif"\n',
    "hdr": 'p f0 w240 h160 v80 n"TIFF"\ni w200 h150 f0 v70 y0 p0 r0 Eev10 n"a.tif"\ni w200 h150 f0 v70 y0.5 p0 r0 Eev12 n"b.tif"\ni w200 h150 f0 v70 y0 p0.5 r0 Eev14 n"c.tif"\n',
    "translation": 'p f0 w260 h180 v90 n"TIFF"\ni w200 h150 f0 v60 y-10 p0 r0 n"a.tif"\ni w200 h150 f0 v60 y12 p2 r1 TrX0.2 TrY-0.05 TrZ0.1 Tpy5 Tpp-3 n"b.tif"\n',
    "masks_and_crops": '''p f2 w800 h400 v360 n"TIFF_m"
i w400 h300 f0 v60 y0 p0 r0 S10,390,20,280 n"with blank in name.tif"
i w400 h300 f3 v=0 y90 p=0 r=0 C5,395,5,295 n"b.tif"
k i0 t0 p"10 10 100 10 100 100 10 100"
k i1 t1 p"200.5 20 300 40.25 250 200"
v Ra0 Rb0
c n0 N1 x100 y200 X110 Y205 t0
''',
    "odd_lines": '''
this line is not a PTO line
p
pf2 w10
i  w100 h50   f0 v45
i w100 h50 f0 v45 j=0 n"x.tif"
z 1 2 3
i\tw7 h8 f0 v30
i w9 h9 f0 v=1 n"ref to second.tif"
''',
}
